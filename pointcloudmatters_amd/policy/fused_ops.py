"""Fused transformer tail ops backed by csrc/drln.hip, and the per-step context they need.

``drln(x, y, norm, dropout)`` = ``norm(x + dropout(y))`` in one kernel each way.  The dropout mask is a
counter-based hash of (seed, call site, element); the seed lives in a device tensor owned by the
training loop (so hipGraph replays draw fresh masks) and the call-site index is a Python counter that
restarts with every forward -- both come from the active :class:`FusedContext`.  Without an active
context (CPU runs, reference-order tests, eager mode) callers fall back to the framework ops.
"""
import contextlib

import torch
from torch.autograd import Function

from .. import _lib

_ACTIVE = None


class FusedContext:
    def __init__(self, device, base_seed=0x5EED):
        self.device = torch.device(device)
        self.seed = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._host = torch.zeros(1, dtype=torch.int64).pin_memory() if self.device.type == "cuda" else torch.zeros(1, dtype=torch.int64)
        self.base_seed = int(base_seed)
        self.site = 0

    def set_step(self, step):
        """Host side, between steps / graph replays: an asynchronous 8-byte copy on the current stream."""
        self._host[0] = self.base_seed + 1000003 * int(step)
        self.seed.copy_(self._host, non_blocking=True)

    def next_site(self):
        self.site += 1
        return self.site


@contextlib.contextmanager
def activate(ctx):
    global _ACTIVE
    prev, _ACTIVE = _ACTIVE, ctx
    if ctx is not None:
        ctx.site = 0
    try:
        yield ctx
    finally:
        _ACTIVE = prev


def current():
    return _ACTIVE


class _DRLN(Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps, p_drop, seed, site):
        L = _lib.load()
        shape = x.shape
        E = shape[-1]
        x2 = x.reshape(-1, E)
        y2 = y.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if not y2.is_contiguous():
            y2 = y2.contiguous()
        R = x2.shape[0]
        dev = x.device
        with torch.cuda.device(dev):
            s = torch.empty_like(x2)
            out = torch.empty_like(x2)
            mean = torch.empty(R, dtype=torch.float32, device=dev)
            rstd = torch.empty(R, dtype=torch.float32, device=dev)
            rc = L.pcm_drln_forward_hip(R, E, 1 if y2.dtype == torch.bfloat16 else 0, x2.data_ptr(), y2.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), float(eps), float(p_drop), seed.data_ptr() if seed is not None else 0,
                                        int(site), s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pcm_drln_forward_hip")
        ctx.save_for_backward(s, mean, rstd, gamma)
        ctx.meta = (shape, y.dtype, float(p_drop), seed, int(site))
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout):
        L = _lib.load()
        s, mean, rstd, gamma = ctx.saved_tensors
        shape, ydtype, p_drop, seed, site = ctx.meta
        R, E = s.shape
        dev = s.device
        d2 = dout.reshape(R, E)
        if d2.dtype != torch.float32 or not d2.is_contiguous():
            d2 = d2.float().contiguous()
        with torch.cuda.device(dev):
            dx = torch.empty_like(s)
            dy = torch.empty(R, E, dtype=ydtype, device=dev)
            partial = torch.empty(L.pcm_drln_blocks(R) * 2 * E, dtype=torch.float32, device=dev)
            dgb = torch.empty(2, E, dtype=torch.float32, device=dev)
            rc = L.pcm_drln_backward_hip(R, E, 1 if ydtype == torch.bfloat16 else 0, d2.data_ptr(), s.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), gamma.data_ptr(), p_drop, seed.data_ptr() if seed is not None else 0, site,
                                         dx.data_ptr(), dy.data_ptr(), partial.data_ptr(), dgb.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pcm_drln_backward_hip")
        return dx.view(shape), dy.view(shape), dgb[0], dgb[1], None, None, None, None


def drln_supported(x, y, norm):
    e = x.shape[-1]
    return (_ACTIVE is not None and x.is_cuda and x.dtype == torch.float32 and y.dtype in (torch.float32, torch.bfloat16)
            and type(norm) is torch.nn.LayerNorm and norm.elementwise_affine and norm.bias is not None
            and e % 256 == 0 and e <= 1024 and x.shape == y.shape)


def drln(x, y, norm, dropout):
    """norm(x + dropout(y)) through the fused kernel; the caller checked ``drln_supported``."""
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    ctx = _ACTIVE
    return _DRLN.apply(x, y, norm.weight, norm.bias, norm.eps, p, ctx.seed if p > 0 else None, ctx.next_site())
