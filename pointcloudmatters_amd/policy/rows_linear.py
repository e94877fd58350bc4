"""``linear_rows(x, weight, bias)`` = ``F.linear`` whose weight gradient is a split-K product.

Every Linear of this path is applied to a tall activation matrix -- all points of the batch for the PointNet / SA
layers (8 192 ... 131 072 rows), all tokens of the batch for the transformer projections (4 120 rows) -- so its weight
gradient dW = dY^T X is a GEMM with a small output (<= 1024 x 512) and a very long reduction.  hipBLASLt serves that
shape with a single-pass kernel on a handful of workgroups (measured on MI355X, bf16: 51 us at 8 192 x 512 x 512,
336 us at 131 072 x 64 x 64, against 10 us / 11 us for the forward GEMM of the same layer).  Here the reduction is cut
into S row blocks -- one batched GEMM producing S partial products, then one fp32 sum over S -- which fills the chip
(19 us and 18 us for the two shapes above) and is at least as accurate: the partials stay fp32 and are summed in a fixed
order by pcm_slab_sum_hip with a single rounding to the gradient dtype.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

MIN_ROWS = 2048        # below this the plain GEMM is as fast
TARGET_CHUNK = 768     # rows per partial product
MAX_SPLITS = 64


def _splits(rows):
    s = max(1, min(MAX_SPLITS, rows // TARGET_CHUNK))
    for cand in range(s, max(1, s // 2), -1):  # prefer an exact divisor close to the target
        if rows % cand == 0:
            return cand
    return s


def weight_grad(go, x, out_dtype, out=None):
    """dW = go^T @ x for go (rows, m), x (rows, k) -> (m, k) in `out_dtype` (written into `out`, a contiguous (m, k)
    tensor or row-slice of a packed gradient, when given)."""
    rows, m = go.shape
    k = x.shape[1]
    if rows < MIN_ROWS or m * k > 1024 * 1024:
        if out is not None and out.dtype == go.dtype and out.is_contiguous():
            return torch.mm(go.t(), x, out=out)  # straight into the (slice of the) packed gradient: no copy kernel
        dw = go.t() @ x
        if out is not None:
            return out.copy_(dw)
        return dw.to(out_dtype)
    s = _splits(rows)
    chunk = rows // s
    main = s * chunk
    a, b = go[:main].view(s, chunk, m).transpose(1, 2), x[:main].view(s, chunk, k)
    # fp32 partial products (bf16 inputs keep their fp32 accumulators), one fixed-order sum, ONE rounding to the output dtype
    part = torch.bmm(a, b, out_dtype=torch.float32) if a.dtype != torch.float32 else torch.bmm(a, b)
    tail = (go[main:].t() @ x[main:]).float() if main < rows else None
    if tail is None and out_dtype in (torch.float32, torch.bfloat16) and (out is None or out.is_contiguous()):
        from .. import _lib

        dw = out if out is not None else torch.empty(m, k, dtype=out_dtype, device=go.device)
        with torch.cuda.device(go.device):
            rc = _lib.load().pcm_slab_sum_hip(s, m * k, part.data_ptr(), int(dw.dtype == torch.bfloat16), dw.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pcm_slab_sum_hip")
        return dw
    dw = part.sum(dim=0)
    if tail is not None:  # fewer than s leftover rows
        dw = dw + tail
    if out is not None:
        return out.copy_(dw)
    return dw.to(out_dtype)


class _LinearRows(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            xc, wc = x.to(dt), weight.to(dt)
            bc = bias.to(dt) if bias is not None else None
        else:
            xc, wc, bc = x, weight, bias
        with torch.autocast("cuda", enabled=False):
            y = F.linear(xc, wc, bc)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, bias.dtype if bias is not None else None, x.shape)
        return y

    @staticmethod
    def backward(ctx, go):
        xc, wc = ctx.saved_tensors
        xdt, wdt, bdt, xshape = ctx.meta
        go2 = go.reshape(-1, go.shape[-1])
        if go2.dtype != xc.dtype:
            go2 = go2.to(xc.dtype)
        if not go2.is_contiguous():
            go2 = go2.contiguous()
        x2 = xc.reshape(-1, xc.shape[-1])
        dx = dw = db = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                dx = (go2 @ wc).view(xshape)
                if dx.dtype != xdt:
                    dx = dx.to(xdt)
            if ctx.needs_input_grad[1]:
                dw = weight_grad(go2, x2 if x2.is_contiguous() else x2.contiguous(), wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                db = go2.sum(dim=0).to(bdt)
        return dx, dw, db


def linear_rows(x, weight, bias=None):
    """Drop-in for F.linear(x, weight, bias) on activations with many rows (any leading shape)."""
    rows = x.numel() // max(1, x.shape[-1])
    if not x.is_cuda or rows < MIN_ROWS or not torch.is_grad_enabled() or not (
            weight.requires_grad or x.requires_grad or (bias is not None and bias.requires_grad)):
        return F.linear(x, weight, bias)
    return _LinearRows.apply(x, weight, bias)
