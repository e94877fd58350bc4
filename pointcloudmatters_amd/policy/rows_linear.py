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
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function
from .._lib import raw_stream as _raw_stream

MIN_ROWS = 2048        # below this the plain GEMM is as fast
TARGET_CHUNK = 768     # rows per partial product
MAX_SPLITS = 64
MAX_PARTIAL_BYTES = 64 << 20  # fp32 partial products of one weight gradient (S x m x k)
DEFER_MAX_BYTES = int(os.environ.get("PCM_DEFER_SLAB_BYTES", 0))  # split-K sums up to this size may join policy/deferred.py's batch
WIDE_OUTPUT = 1024 * 1024      # m * k above which the fixed cost of summing the partials (20 us at 3584 x 512) ...
WIDE_MIN_ROWS = 8192           # ... only pays for this many rows (4120 rows: 44 us unsplit vs 48 + 20 split; 16408: 216 vs 95)


def _splits(rows, s_max=MAX_SPLITS):
    """Number of row blocks (at most `s_max`).  Powers of two first: measured on MI355X (bf16, device time), hipBLASLt's batched kernels for
    S = 4 / 8 / 16 are up to 2x faster than for S = 5 at the same total work (4120 x 1024 x 512: 24 us vs 51 us), so the
    largest power of two that divides the rows and leaves >= TARGET_CHUNK rows per block wins; any other divisor close to
    the target is the fallback."""
    s = MAX_SPLITS
    while s >= 2:
        if s <= s_max and rows % s == 0 and rows // s >= TARGET_CHUNK:
            return s
        s //= 2
    s = max(1, min(s_max, rows // TARGET_CHUNK))
    for cand in range(s, max(1, s // 2), -1):  # an exact divisor close to the target
        if rows % cand == 0:
            return cand
    return s


class _SideQueue:
    """Weight gradients off the critical path.  In backward the chain of INPUT gradients (dX of layer l feeds layer
    l-1) is the critical path; each layer's dW = dY^T X is a leaf of the dependency graph that nothing reads until the
    gradients are handed to the optimizer.  These products are small (a 512 x 512 output fills a quarter of the chip),
    so with the trainer's consent they are launched on a second HIP stream: in a captured step they become parallel
    branches of the hipGraph and run beside the next layers' dX GEMMs / attention kernels instead of between them.

    Safety: the side stream waits for the main stream before each product (its operands are ready); the operands are
    kept alive here until ``join`` (the caching allocator may otherwise recycle them for main-stream work while the
    side product is still pending); the result is allocated before the switch and must not be read by main-stream
    work before ``join`` -- the trainer joins before it collects the gradients, and callers pass ``side=True`` only
    for weights whose gradient goes straight to the optimizer (leaf or bf16-shadow parameters)."""

    def __init__(self):
        self.active = False
        self.stream = None
        self.held = []
        self.pending = False
        self.min_rows = 0

    def stream_for(self, device):
        if self.stream is None or self.stream.device != device:
            self.stream = torch.cuda.Stream(device=device)
        return self.stream

    def join(self):
        if self.pending:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
            self.pending = False
        self.held.clear()


SIDE = _SideQueue()


def join_side():
    SIDE.join()


def goes_to_optimizer(weight):
    """True when the gradient of `weight` is handed to the optimizer without passing through another kernel: a leaf
    parameter, or the bf16 mirror of one (bc/trainer.py _ShadowParam)."""
    fn = weight.grad_fn
    # _pcm_defer_ok: set by nodes that only pass the gradient on (views) or copy it inside the deferral window themselves
    # (transformer.split_packed / pack_rows)
    return fn is None or type(fn).__name__ == "_ShadowParamBackward" or bool(getattr(weight, "_pcm_defer_ok", False))


def weight_grad(go, x, out_dtype, out=None, side=False, defer=False, tag=None):
    """dW = go^T @ x for go (rows, m), x (rows, k) -> (m, k) in `out_dtype` (written into `out`, a contiguous (m, k)
    tensor or row-slice of a packed gradient, when given).  ``side``: may run on the side stream (see _SideQueue).
    ``defer``: the closing sum of a split product may stay pending until policy/deferred.flush (the caller checked
    deferred.clear: nothing reads dW before the gradient hand-off)."""
    if side and SIDE.active and go.is_cuda and go.shape[0] >= SIDE.min_rows:
        main = torch.cuda.current_stream(go.device)
        st = SIDE.stream_for(go.device)
        if out is None:
            out = torch.empty(go.shape[1], x.shape[1], dtype=out_dtype, device=go.device)  # main stream's allocation
        st.wait_stream(main)
        with torch.cuda.stream(st):
            _weight_grad(go, x, out_dtype, out)
        SIDE.held.append((go, x))
        SIDE.pending = True
        return out
    return _weight_grad(go, x, out_dtype, out, defer, tag)


def _weight_grad(go, x, out_dtype, out=None, defer=False, tag=None):
    rows, m = go.shape
    k = x.shape[1]
    # wide outputs (the decoder's 7-layer key / value projection: 3584 x 512) are split too, as long as the fp32
    # partials stay small: unsplit, hipBLASLt runs that product on 65 workgroups (216 us at 16408 rows)
    s_max = min(MAX_SPLITS, MAX_PARTIAL_BYTES // (m * k * 4))
    if defer:
        # one batched product per (site, shape) at the end of the backward stage -- also for the encoder's 4120 rows: four
        # layers' unsplit products in one launch fill the chip (31 us) where four split products + closing sums take 76
        from . import deferred

        dw = deferred.push_wgrad(go, x, out_dtype, out, tag)
        if dw is not None:
            return dw
    if rows < MIN_ROWS or s_max < 2 or (m * k > WIDE_OUTPUT and rows < WIDE_MIN_ROWS):
        if out is not None and out.dtype == go.dtype and out.is_contiguous():
            return torch.mm(go.t(), x, out=out)  # straight into the (slice of the) packed gradient: no copy kernel
        dw = go.t() @ x
        if out is not None:
            return out.copy_(dw)
        return dw.to(out_dtype)
    s = _splits(rows, s_max)
    chunk = rows // s
    main = s * chunk
    a, b = go[:main].view(s, chunk, m).transpose(1, 2), x[:main].view(s, chunk, k)
    # fp32 partial products (bf16 inputs keep their fp32 accumulators), one fixed-order sum, ONE rounding to the output dtype
    part = torch.bmm(a, b, out_dtype=torch.float32) if a.dtype != torch.float32 else torch.bmm(a, b)
    tail = (go[main:].t() @ x[main:]).float() if main < rows else None
    if tail is None and out_dtype in (torch.float32, torch.bfloat16) and (out is None or out.is_contiguous()):
        from .. import _lib

        dw = out if out is not None else torch.empty(m, k, dtype=out_dtype, device=go.device)
        # small sums only: a large one reads its partials out of the caches when it runs right behind the product that wrote
        # them (32 MiB in 7 us); at the end of the stage the same sum comes from HBM (measured at C2: 107 -> 250 us in total)
        if defer and part.numel() * 4 <= DEFER_MAX_BYTES:
            from . import deferred

            if deferred.push(part, s, m * k, **({"out_bf16": dw} if dw.dtype == torch.bfloat16 else {"out_f32": dw})):
                return dw if out is not None else deferred.handout(dw)
        with torch.cuda.device(go.device):
            rc = _lib.load().pcm_slab_sum_hip(s, m * k, part.data_ptr(), int(dw.dtype == torch.bfloat16), dw.data_ptr(),
                                              _raw_stream())
        _lib.check(rc, "pcm_slab_sum_hip")
        return dw
    dw = part.sum(dim=0)
    if tail is not None:  # fewer than s leftover rows
        dw = dw + tail
    if out is not None:
        return out.copy_(dw)
    return dw.to(out_dtype)


def bias_grad(go, out_dtype, defer=False):
    """Column sums of go (rows, C) = the bias gradient of a linear layer, by csrc/tokens.hip's two-stage column sum (fp32
    partial rows in a fixed order, one rounding) instead of the framework's reduction (a memset + a kernel twice as slow
    at 800 x 512); the closing stage joins policy/deferred.py's batch when `defer`.  Falls back to go.sum(0)."""
    rows, C = go.shape
    if not go.is_cuda or rows == 0 or C % 4 or go.dtype not in (torch.bfloat16, torch.float32) \
            or out_dtype not in (torch.bfloat16, torch.float32) or not go.is_contiguous():
        # fp32 accumulation through the framework's fp32 reduction: its bf16 column-sum kernel is one of the kernels that return wrong
        # values beside another stream's MFMA work on this stack (packed-fp32 hazard, DESIGN.md section 2: (816, 512).sum(0) in bf16
        # wrong in 28 of 50 runs beside a GEMM graph, the fp32 reductions in 0)
        return (go.float() if go.is_cuda and go.dtype == torch.bfloat16 else go).sum(dim=0).to(out_dtype)
    from .. import _lib
    from . import deferred

    L = _lib.load()
    chunk = min(C, 1024)
    pieces = [(c0, chunk) for c0 in range(0, C - C % chunk, chunk)]
    if C % chunk:
        pieces.append((C - C % chunk, C % chunk))
    db = torch.empty(C, dtype=out_dtype, device=go.device)
    any_pending = False
    es, os_ = go.element_size(), db.element_size()
    st = _raw_stream()
    with torch.cuda.device(go.device):
        i = 0
        while i < len(pieces):
            c0, w = pieces[i]
            nt = 1
            while nt < 3 and i + nt < len(pieces) and pieces[i + nt][1] == w:
                nt += 1
            slots = L.pcm_colsum_slots(rows, w)
            partial = torch.empty(slots * nt * w, dtype=torch.float32, device=go.device)
            ptr = [go.data_ptr() + (c0 + j * w) * es if j < nt else 0 for j in range(3)]
            out = db[c0: c0 + nt * w]
            pending = defer and deferred.push(partial, slots, nt * w, **({"out_bf16": out} if out_dtype == torch.bfloat16 else {"out_f32": out}))
            any_pending = any_pending or bool(pending)
            if pending and deferred.push_colsum(rows, w, ptr[:nt], [C] * nt, go.dtype, partial, go):
                i += nt  # both stages ride the batch
                continue
            rc = L.pcm_colsum_hip(rows, w, nt, int(go.dtype == torch.bfloat16), ptr[0], C, ptr[1], C, ptr[2], C, partial.data_ptr(),
                                  int(out_dtype == torch.bfloat16), 0 if pending else db.data_ptr() + c0 * os_, st)
            _lib.check(rc, "pcm_colsum_hip")
            i += nt
    return deferred.handout(db) if any_pending else db


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
        ctx.side_ok = goes_to_optimizer(weight)
        from . import deferred

        ctx.defer = deferred.targets(weight)
        ctx.defer_b = deferred.targets(bias) if bias is not None else (False, [])
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
        from . import deferred

        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                dx = (go2 @ wc).view(xshape)
                if dx.dtype != xdt:
                    dx = dx.to(xdt)
            if ctx.needs_input_grad[1]:
                from . import deferred

                dw = weight_grad(go2, x2 if x2.is_contiguous() else x2.contiguous(), wdt, side=ctx.side_ok,
                                 defer=deferred.clear(*ctx.defer), tag="linear_rows")
            if bdt is not None and ctx.needs_input_grad[2]:
                db = bias_grad(go2, bdt, defer=deferred.clear(*ctx.defer_b))
        return dx, dw, db


def linear_rows(x, weight, bias=None):
    """Drop-in for F.linear(x, weight, bias) on activations with many rows (any leading shape)."""
    rows = x.numel() // max(1, x.shape[-1])
    if not x.is_cuda or not torch.is_grad_enabled() or not (
            weight.requires_grad or x.requires_grad or (bias is not None and bias.requires_grad)):
        return F.linear(x, weight, bias)
    if rows < MIN_ROWS:
        # short activations: the plain products are as fast, but under bf16 autocast the framework would reduce the bias gradient with
        # its bf16 column-sum kernel (see bias_grad): keep the node, whose backward sums through csrc/tokens.hip, when there is a bias
        # gradient to form
        safe = bias is not None and bias.requires_grad and rows > 0 and torch.is_autocast_enabled("cuda") \
            and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if not safe:
            return F.linear(x, weight, bias)
    return _LinearRows.apply(x, weight, bias)


class RowsLinear(torch.nn.Linear):
    """nn.Linear whose forward goes through ``linear_rows``: identical results and state-dict keys; in training on the GPU under bf16
    autocast the backward forms the bias gradient with the library's column sums (``bias_grad``) instead of the framework's bf16
    reduction kernel -- one of the kernels that miscompute beside another stream's MFMA work on this stack."""

    def forward(self, x):
        return linear_rows(x, self.weight, self.bias)
