"""Channels-last ops of the Diffusion-Policy U-Net backed by csrc/gnmish.hip.

Activations are (B, T, C) throughout: that is what an im2col GEMM consumes and produces, so no transpose / contiguous
copy is ever made.  On HIP tensors the ops below are single launches of the C-ABI kernels; on host tensors (golden
fixtures, CPU parity tests) the same maths goes through framework ops in the reference's order.

  conv1d_cl(x, conv)            nn.Conv1d          as  pcm_im2col_cl + one hipBLASLt GEMM (bias fused)
  conv_transpose1d_cl(x, conv)  nn.ConvTranspose1d(k=4, s=2, p=1)  as one GEMM + overlap-add
  gn_mish_cl(x, norm, ...)      Mish(GroupNorm(x)) [FiLM] [+ residual]  as pcm_gn_mish (one launch each way)
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from .rows_linear import linear_rows
from .._lib import raw_stream as _raw_stream


def _stream():
    return _raw_stream()


def _bf(t):
    return int(t.dtype == torch.bfloat16)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _Im2colCL(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, out_dtype):
        L = _lib.load()
        b, t, c = x.shape
        lout = (t + 2 * pad - k) // stride + 1
        x = x.contiguous()
        with torch.cuda.device(x.device):
            cols = torch.empty(b * lout, c * k, dtype=out_dtype, device=x.device)
            rc = L.pcm_im2col_cl_hip(b, t, c, k, stride, pad, _bf(x), x.data_ptr(), _bf(cols), cols.data_ptr(), _stream())
        _lib.check(rc, "pcm_im2col_cl_hip")
        ctx.meta = (b, t, c, k, stride, pad, x.dtype)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        L = _lib.load()
        b, t, c, k, stride, pad, xdtype = ctx.meta
        dcols = dcols.contiguous()
        with torch.cuda.device(dcols.device):
            dx = torch.empty(b, t, c, dtype=xdtype, device=dcols.device)
            rc = L.pcm_col2im_cl_hip(b, t, c, k, stride, pad, _bf(dcols), dcols.data_ptr(), _bf(dx), dx.data_ptr(), _stream())
        _lib.check(rc, "pcm_col2im_cl_hip")
        return dx, None, None, None, None


def _hip_ok(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)


def im2col_cl(x, k, stride, pad):
    """(B, T, C) -> (B*L_out, C*K) with column order (c, k); bf16 under autocast."""
    if _hip_ok(x):
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if out_dtype not in (torch.float32, torch.bfloat16):
            out_dtype = x.dtype
        return _Im2colCL.apply(x, k, stride, pad, out_dtype)
    b, t, c = x.shape
    xp = F.pad(x, (0, 0, pad, pad)) if pad else x
    cols = xp.unfold(1, k, stride)  # (B, L_out, C, K) view
    return cols.reshape(b * cols.shape[1], c * k)


def conv1d_cl(x, conv, with_bias=True):
    """nn.Conv1d on channels-last activations: (B, T, C_in) -> (B, L_out, C_out).  with_bias=False leaves the bias to the
    consumer (gn_mish_cl(conv_bias=...))."""
    k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    b, t, cin = x.shape
    w = conv.weight  # (C_out, C_in, K)
    bias = conv.bias if with_bias else None
    if k == 1 and stride == 1 and pad == 0:
        return linear_rows(x, w[:, :, 0], bias)
    cols = im2col_cl(x, k, stride, pad)
    y = linear_rows(cols, w.reshape(w.shape[0], cin * k), bias)
    return y.view(b, -1, w.shape[0])


def conv_transpose1d_cl(x, conv):
    """nn.ConvTranspose1d(C, C, 4, 2, 1) (Upsample1d) on channels-last activations: (B, L, C) -> (B, 2L, C_out).
    One GEMM gives the (B, L, C_out, 4) taps; output row o = 2l + k - 1 sums taps (l, k) -- two per row."""
    assert conv.kernel_size[0] == 4 and conv.stride[0] == 2 and conv.padding[0] == 1 and conv.output_padding[0] == 0
    b, l, cin = x.shape
    w = conv.weight  # (C_in, C_out, 4)
    cout = w.shape[1]
    taps = F.linear(x.reshape(b * l, cin), w.reshape(cin, cout * 4).t()).view(b, l, cout, 4)
    # rows of a (L+1, 2) grid: taps 0,1 of step l land on grid row l, taps 2,3 on grid row l+1
    lo = F.pad(taps[..., 0:2], (0, 0, 0, 0, 0, 1))  # (B, L+1, C_out, 2)
    hi = F.pad(taps[..., 2:4], (0, 0, 0, 0, 1, 0))
    full = (lo + hi).permute(0, 1, 3, 2).reshape(b, 2 * l + 2, cout)
    y = full[:, 1:-1]
    return y + conv.bias if conv.bias is not None else y


class _GNMish(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, film, res, conv_bias, groups, eps, film_mode):
        L = _lib.load()
        b, t, c = x.shape
        x = x.contiguous()
        film_c = film.contiguous() if film is not None else None
        res_c = res.contiguous() if res is not None else None
        cb = conv_bias.float().contiguous() if conv_bias is not None else None
        dev = x.device
        with torch.cuda.device(dev):
            y = torch.empty(b, t, c, dtype=torch.float32, device=dev)
            stats = torch.empty(2, b * groups, dtype=torch.float32, device=dev)
            rc = L.pcm_gn_mish_forward_hip(b, t, c, groups, _bf(x), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                                           int(film_mode), _bf(film_c) if film_c is not None else 0, _ptr(film_c),
                                           _bf(res_c) if res_c is not None else 0, _ptr(res_c), _ptr(cb), y.data_ptr(),
                                           stats[0].data_ptr(), stats[1].data_ptr(), _stream())
        _lib.check(rc, "pcm_gn_mish_forward_hip")
        ctx.save_for_backward(x, gamma, beta, film_c, cb, stats)
        ctx.meta = (groups, int(film_mode), film.dtype if film is not None else None, film.shape if film is not None else None,
                    res.dtype if res is not None else None, conv_bias.dtype if conv_bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.load()
        x, gamma, beta, film, cb, stats = ctx.saved_tensors
        groups, film_mode, film_dtype, film_shape, res_dtype, cb_dtype = ctx.meta
        b, t, c = x.shape
        dy = dy.contiguous().float()
        dev = x.device
        with torch.cuda.device(dev):
            dx = torch.empty_like(x)
            dgb = torch.empty(b, 3, c, dtype=torch.float32, device=dev)
            dfilm = torch.empty(film_shape, dtype=torch.float32, device=dev) if film is not None else None
            rc = L.pcm_gn_mish_backward_hip(b, t, c, groups, _bf(x), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            stats[0].data_ptr(), stats[1].data_ptr(), film_mode,
                                            _bf(film) if film is not None else 0, _ptr(film), _ptr(cb), dy.data_ptr(), dx.data_ptr(),
                                            dgb.data_ptr(), _ptr(dfilm), _stream())
        _lib.check(rc, "pcm_gn_mish_backward_hip")
        dgb = dgb.sum(dim=0)  # (3, C): dgamma | dbeta | d(conv bias)
        if dfilm is not None and film_dtype != torch.float32:
            dfilm = dfilm.to(film_dtype)
        dres = None
        if res_dtype is not None:
            dres = dy if res_dtype == torch.float32 else dy.to(res_dtype)
        dcb = None
        if cb_dtype is not None:
            dcb = dgb[2] if cb_dtype == torch.float32 else dgb[2].to(cb_dtype)
        return dx, dgb[0], dgb[1], dfilm, dres, dcb, None, None, None


def gn_mish_supported(x, norm):
    if not (_hip_ok(x) and x.dim() == 3 and type(norm) is torch.nn.GroupNorm and norm.affine
            and norm.weight.dtype == torch.float32):
        return False
    return bool(_lib.load().pcm_gn_mish_supported(int(x.shape[1]), int(x.shape[2]), int(norm.num_groups)))


def gn_mish_cl(x, norm, film=None, film_mode=0, res=None, conv_bias=None):
    """x (B, T, C) -> mish(GroupNorm(x [+ conv_bias])), then FiLM (film_mode 1: film (B, 2C) = scale | bias; 2: film (B, C) =
    bias), then ``+ res``.  fp32 output (autocast runs group_norm in fp32 too).  `conv_bias` (C): the bias of the
    convolution that produced x, left out of its GEMM and added here instead, so that its gradient falls out of the same
    backward launch (no separate reduction over B*T rows)."""
    if gn_mish_supported(x, norm) and (film is None or film.dtype in (torch.float32, torch.bfloat16)) \
            and (res is None or res.dtype in (torch.float32, torch.bfloat16)):
        return _GNMish.apply(x, norm.weight, norm.bias, film, res, conv_bias, norm.num_groups, norm.eps,
                             film_mode if film is not None else 0)
    if conv_bias is not None:
        x = x + conv_bias.to(x.dtype)
    y = F.mish(F.group_norm(x.transpose(1, 2), norm.num_groups, norm.weight, norm.bias, norm.eps)).transpose(1, 2)
    if film is not None:
        c = x.shape[2]
        if film_mode == 1:
            y = film[:, None, :c] * y + film[:, None, c:]
        else:
            y = y + film[:, None, :]
    if res is not None:
        y = y + res
    return y
