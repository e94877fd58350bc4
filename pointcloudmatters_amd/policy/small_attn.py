"""Attention cores on the matrix cores as autograd functions: csrc/attn_small.hip for short query sets, csrc/attn_flash.hip
for long ones (the encoder's self-attention over the point tokens).

``small_attention(q, k, v, key_padding_mask, heads, dropout_p)`` == softmax(q k^T / sqrt(d) + mask) (dropout) v per head
for q (B, L, E), k / v (B, S, E) in bf16 with E = heads * 64 (the policy uses it for L <= 128, S <= 4096); returns (B, L, E) -- already in the layout
the output projection wants (the SDPA path needs a transpose copy).  Used by policy/transformer.attention for the CVAE
encoder and the decoder; query sets longer than MAX_QUERIES (the encoder self-attention over S = M + 3 tokens) go to the
same interface backed by csrc/attn_flash.hip (PCM_ATTN_LONG=sdpa keeps them on the framework's kernel, for A/B runs).
"""
import math
import os

import torch
from torch.autograd import Function

from .. import _lib
from . import deferred
from .._lib import raw_stream as _raw_stream


# The kernels handle any length (tests go to 515 x 515), but they win only for short query sets: measured on MI355X at
# B*H = 64, 515 x 515 the forward ties the framework's flash kernel (45 us) and the backward loses (233 vs 128 us), so
# the policy routes only <= 128 queries here (CVAE encoder, decoder self- and cross-attention).
MAX_QUERIES = 128
MAX_KEYS = 4096  # cross-attention over the REF memory (2051 tokens): forward 28 us vs 61 us for the flash kernel, backward on par


LONG_IMPL = os.environ.get("PCM_ATTN_LONG", "flash")  # "flash" (csrc/attn_flash.hip) | "sdpa" (framework kernel)
FLASH_FROM = int(os.environ.get("PCM_ATTN_FLASH_FROM", "129"))  # query count from which attn_flash.hip takes over from attn_small.hip


def _st(t):
    return t.stride(0), t.stride(1)


class _SmallAttn(Function):
    @staticmethod
    def forward(ctx, q, k, v, kpm, heads, p_drop, seed, site, slots=(None, None)):
        L_ = _lib.load()
        B, L, E = q.shape
        S = k.shape[1]
        q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
        dev = q.device
        with torch.cuda.device(dev):
            out = deferred.take((B, L, E), torch.bfloat16, dev, "attn.out")
            lse = torch.empty(B, heads, L, dtype=torch.float32, device=dev)
            use_flash = L >= FLASH_FROM
            fwd = L_.pcm_attn_flash_forward_hip if use_flash else L_.pcm_attn_small_forward_hip
            rc = fwd(B, heads, L, S, q.data_ptr(), *_st(q), k.data_ptr(), *_st(k), v.data_ptr(), *_st(v),
                     kpm.data_ptr() if kpm is not None else 0, 1.0 / math.sqrt(E // heads), float(p_drop),
                     seed.data_ptr() if seed is not None else 0, int(site), out.data_ptr(),
                     lse.data_ptr(), _raw_stream())
        _lib.check(rc, "pcm_attn_flash_forward_hip" if use_flash else "pcm_attn_small_forward_hip")
        ctx.save_for_backward(q, k, v, out, lse, kpm)
        ctx.meta = (heads, float(p_drop), seed, int(site), use_flash)
        ctx.slots = slots
        return out

    @staticmethod
    def backward(ctx, dout):
        L_ = _lib.load()
        q, k, v, out, lse, kpm = ctx.saved_tensors
        heads, p_drop, seed, site, use_flash = ctx.meta
        B, L, E = q.shape
        S = k.shape[1]
        dout = dout.to(torch.bfloat16).contiguous()
        dev = q.device
        with torch.cuda.device(dev):
            from . import fused_ops

            if L == S and ctx.slots[1] is None and B * L < fused_ops.INPROJ_MERGE_ROWS:
                # short self-attention: dq | dk | dv side by side -- the packed in-projection then gets its three input
                # gradients from ONE batched product (fused_ops._SelfAttnInProj.backward)
                dqkv = deferred.take((B, L, 3, E), torch.bfloat16, dev, "attn.dqkv")
                dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
            elif L == S:  # self-attention: dq | dk side by side, the layout the packed in-projection consumes without a copy
                dqk = deferred.take((B, L, 2, E), torch.bfloat16, dev, "attn.dqk")
                dq, dk = dqk[:, :, 0], dqk[:, :, 1]
                dv = _grad_buffer(ctx.slots[1], B, S, E, dev)
            else:
                dq = deferred.take((B, L, E), torch.bfloat16, dev, "attn.dq")
                dk = _grad_buffer(ctx.slots[0], B, S, E, dev)
                dv = _grad_buffer(ctx.slots[1], B, S, E, dev)
            head = (B, heads, L, S, q.data_ptr(), *_st(q), k.data_ptr(), *_st(k), v.data_ptr(), *_st(v),
                    kpm.data_ptr() if kpm is not None else 0, 1.0 / math.sqrt(E // heads), p_drop,
                    seed.data_ptr() if seed is not None else 0, site, out.data_ptr(), dout.data_ptr(), lse.data_ptr())
            tail = (dq.data_ptr(), *_st(dq), dk.data_ptr(), *_st(dk), dv.data_ptr(), *_st(dv), _raw_stream())
            if use_flash:
                delta = torch.empty(B, heads, L, dtype=torch.float32, device=dev)
                rc = L_.pcm_attn_flash_backward_hip(*head, delta.data_ptr(), *tail)
            else:
                rc = L_.pcm_attn_small_backward_hip(*head, *tail)
        _lib.check(rc, "pcm_attn_flash_backward_hip" if use_flash else "pcm_attn_small_backward_hip")
        return dq, dk, dv, None, None, None, None, None, None


def _grad_buffer(slot, B, S, E, dev):
    """Where dK / dV go: the caller's shared gradient buffer (transformer.GradArena, written through its strides) when
    the tensor came out of a `shared_unbind`, else a fresh (B, S, E) tensor."""
    if slot is not None:
        arena, l = slot
        if arena.shape == (B, S, E) and arena.dtype == torch.bfloat16:
            return arena.slot(l)
    return deferred.take((B, S, E), torch.bfloat16, dev, "attn.dkv")


def supported(q, k, v, heads, dropout_p=0.0):
    """True when this call goes to the hand-written kernels: <= MAX_QUERIES queries -> attn_small, more -> attn_flash."""
    from . import fused_ops

    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.dim() == 3):
        return False
    e = q.shape[-1]
    if e % heads or e // heads != 64 or q.shape[1] < 1 or k.shape[1] < 1:
        return False
    if q.shape[1] > MAX_QUERIES:
        if LONG_IMPL != "flash":
            return False
    elif k.shape[1] > MAX_KEYS:
        return False
    if dropout_p > 0 and fused_ops.current() is None:
        return False  # the dropout seed lives in the training loop's FusedContext
    for t in (q, k, v):
        if t.stride(-1) == 1 and (t.stride(0) % 8 or t.stride(1) % 8):
            return False
    return True


def small_attention(q, k, v, key_padding_mask, heads, dropout_p=0.0):
    from . import fused_ops

    kpm = None
    if key_padding_mask is not None:
        kpm = key_padding_mask.contiguous()
        kpm = kpm.view(torch.uint8) if kpm.dtype == torch.bool else kpm.to(torch.uint8)
    ctx = fused_ops.current()
    seed, site = (ctx.seed, ctx.next_site()) if dropout_p > 0 else (None, 0)
    slots = (getattr(k, "_pcm_grad_slot", None), getattr(v, "_pcm_grad_slot", None))
    return _SmallAttn.apply(q, k, v, kpm, heads, dropout_p, seed, site, slots)
