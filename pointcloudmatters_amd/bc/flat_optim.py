"""AdamW + global-norm clipping over one flat HBM buffer, driven by csrc/optim.hip.

All trainable parameters of the policy are re-homed as views of one fp32 buffer (their gradients as
views of a second one), so that
  * the optimizer tail of a training step is two streaming kernels instead of ~10 multi-tensor
    passes (clip_grad_norm_'s norms + scale, AdamW's foreach groups),
  * data parallelism needs exactly ONE gradient all-reduce over RCCL on a contiguous buffer,
  * every step-dependent scalar lives in a device array -> the whole step is hipGraph-capturable.
Numerics: torch.optim.AdamW's update rule element for element (tests compare against it).
"""
import math

import torch

from .. import _lib
from .._host import PinnedRing
from .._lib import raw_stream as _raw_stream

_ALIGN = 64  # floats: every parameter view starts on a 256-byte boundary


class FlatAdamW:
    H_LR, H_BETA1, H_BETA2, H_EPS, H_WD, H_BC1, H_BC2_SQRT, H_MAX_NORM, H_GRAD_SCALE, H_COUNT = range(10)

    def __init__(self, params, schedule, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=0.0, grad_scale=1.0):
        """`params`: an iterable of parameters, or a list of {"params": [...], "weight_decay": wd}
        groups (timm-style no-decay group for biases / norm weights, src/utils/optimizer.py:152-170).
        Groups are laid out back to back; the norm runs over everything, Adam once per group."""
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [([p for p in g["params"] if p.requires_grad], float(g.get("weight_decay", weight_decay))) for g in params]
        else:
            groups = [([p for p in params if p.requires_grad], float(weight_decay))]
        groups = [g for g in groups if g[0]]
        params = [p for g in groups for p in g[0]]
        assert params, "no trainable parameters"
        dev = params[0].device
        if not _lib.on_hip(dev):
            raise _lib.PointopsLibraryError("FlatAdamW runs on the HIP device only (CPU runs use torch.optim.AdamW)")
        assert all(p.dtype == torch.float32 and p.device == dev for p in params)
        self.lib = _lib.load()
        self.params = params
        self.schedule = schedule
        self.beta1, self.beta2, self.eps, self.weight_decay = betas[0], betas[1], eps, weight_decay
        self.max_norm, self.grad_scale = float(max_norm or 0.0), float(grad_scale)
        offs, total, self.segments = [], 0, []
        for gp, wd in groups:
            start = total
            for p in gp:
                offs.append(total)
                total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            self.segments.append((start, total - start, wd))
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_p[o : o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_g[o : o + p.numel()].view_as(p)
        self.offsets = offs
        self.g_views = [p.grad for p in params]
        # bf16 mirror of the weights (written by the Adam kernel) + gradient hand-off for shadowed params
        self.flat_p_bf16 = None
        self.shadow = [None] * len(params)
        self._stash = [None] * len(params)
        self.collect_mode = False
        self.hyper = torch.zeros(len(self.segments), self.H_COUNT, dtype=torch.float32, device=dev)
        self._hyper_ring = PinnedRing((len(self.segments), self.H_COUNT), torch.float32, dev)
        self.partials = torch.zeros(self.lib.pcm_optim_partials_capacity(), dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.last_lr = None

    # ---- host side, between graph replays ------------------------------------------------------
    def prepare_step(self):
        """Write the hyper-parameters of the NEXT optimizer step (1-based t = step_count + 1) into
        the device array: an asynchronous 40-byte copy on the current stream."""
        lr, mom = self.schedule.at(self.step_count) if self.schedule is not None else (self._fixed_lr, None)
        beta1 = self.beta1 if mom is None else mom
        t = self.step_count + 1
        host = self._hyper_ring.next()
        for gi, (_, _, wd) in enumerate(self.segments):
            h = host[gi]
            h[self.H_LR], h[self.H_BETA1], h[self.H_BETA2], h[self.H_EPS], h[self.H_WD] = lr, beta1, self.beta2, self.eps, wd
            h[self.H_BC1] = 1.0 - beta1 ** t
            h[self.H_BC2_SQRT] = math.sqrt(1.0 - self.beta2 ** t)
            h[self.H_MAX_NORM], h[self.H_GRAD_SCALE] = self.max_norm, self.grad_scale
        self._hyper_ring.push(self.hyper)  # every group's own row (weight decay differs between groups)
        self.last_lr = lr
        self.step_count += 1

    # ---- bf16 weight mirror / gradient collection ("collect" mode) -------------------------------
    def enable_bf16_mirror(self, shadowed):
        """`shadowed`: set of parameter ids whose consumers run in bf16 (Linear / attention / conv
        weights under autocast).  Their bf16 views are refreshed by the Adam kernel itself, so no
        per-weight cast kernel runs in forward, and their bf16 gradients are handed over through
        ``stash_grad`` and converted into the flat fp32 buffer by one multi-tensor copy."""
        self.flat_p_bf16 = self.flat_p.to(torch.bfloat16)
        for k, (p, o) in enumerate(zip(self.params, self.offsets)):
            if id(p) in shadowed:
                self.shadow[k] = self.flat_p_bf16[o : o + p.numel()].view_as(p)
        self.collect_mode = True
        for p in self.params:
            p.grad = None

    def stash_grad(self, k, g):
        if self._stash[k] is None:
            self._stash[k] = g
            return
        from ..policy import deferred, rows_linear

        rows_linear.join_side()  # a second gradient for the same weight: the sum reads both, possibly side-stream products
        deferred.flush()         # ... or pending closing reductions
        self._stash[k] = self._stash[k] + g

    def collect(self, first, subset=None):
        """Move this micro-batch's gradients into the flat buffer: one multi-tensor copy (first
        micro-batch of an accumulation window) or add per source dtype, instead of one accumulate
        kernel per parameter.  `subset` (ascending parameter indices) restricts the hand-off to those parameters (the
        hybrid trainer collects the captured half and the eager half of a step separately)."""
        by_dtype = {}
        jobs, alive = [], []  # pcm_xfer_batch_hip jobs (dst, src, numel, kind); `alive` keeps the sources until the launch is enqueued
        zero_from = zero_to = None  # run of adjacent gradient slots without a gradient this step: one fill for the run
        fast_ok = self.flat_g.is_cuda
        gbase = self.flat_g.data_ptr()

        def zero_run(a, b):
            if fast_ok:
                jobs.append((gbase + 4 * a, 0, b - a, _lib.XFER_ZERO))
            else:
                self.flat_g[a:b].zero_()

        for k in (range(len(self.params)) if subset is None else subset):
            p = self.params[k]
            g = self._stash[k] if self.shadow[k] is not None else p.grad
            if g is None:
                if first:
                    o = self.offsets[k]
                    if zero_to == o:
                        zero_to = o + p.numel()
                    else:
                        if zero_from is not None:
                            zero_run(zero_from, zero_to)
                        zero_from, zero_to = o, o + p.numel()
                continue
            if fast_ok and g.is_contiguous() and g.device == self.flat_g.device and g.dtype in (torch.bfloat16, torch.float32) \
                    and g.numel() == p.numel():
                bf = g.dtype == torch.bfloat16
                kind = (_lib.XFER_SET_BF16 if bf else _lib.XFER_SET_F32) if first else (_lib.XFER_ADD_BF16 if bf else _lib.XFER_ADD_F32)
                jobs.append((self.g_views[k].data_ptr(), g.data_ptr(), g.numel(), kind))
                alive.append(g)
            else:
                d, s_ = by_dtype.setdefault(g.dtype, ([], []))
                d.append(self.g_views[k])
                s_.append(g)
            self._stash[k] = None
            p.grad = None
        if zero_from is not None:
            zero_run(zero_from, zero_to)
        if jobs:
            # ONE table-driven launch per 96 tensors (csrc/optim.hip): zero runs, bf16 -> fp32 and fp32 copies (first micro-batch of an
            # accumulation window) or adds (the later ones; the bf16 value is widened exactly, same sums as a cast-then-add)
            with torch.cuda.device(self.flat_g.device):
                _lib.xfer_batch(jobs)
            del alive
        for dsts, srcs in by_dtype.values():  # anything the kernel does not take (non-contiguous gradients)
            if first:
                torch._foreach_copy_(dsts, srcs)
            elif srcs[0].dtype == torch.float32:
                torch._foreach_add_(dsts, srcs)
            else:
                torch._foreach_add_(dsts, [s_.to(torch.float32) for s_ in srcs])

    # ---- device side (capturable) -----------------------------------------------------------------
    def zero_grad(self):
        if not self.collect_mode:
            self.flat_g.zero_()

    def launch_step(self):
        """Enqueue norm + AdamW on the current stream.  No host reads: safe inside graph capture."""
        st = _raw_stream()
        lib = self.lib
        import ctypes

        npart = ctypes.c_int(0)
        rc = lib.pcm_grad_sumsq_hip(self.numel, self.flat_g.data_ptr(), self.partials.data_ptr(), ctypes.addressof(npart), st)
        _lib.check(rc, "pcm_grad_sumsq_hip")
        for gi, (start, count, _) in enumerate(self.segments):
            o = start * 4
            rc = lib.pcm_adamw_flat_hip(count, self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o,
                                        self.exp_avg.data_ptr() + o, self.exp_avg_sq.data_ptr() + o,
                                        self.hyper.data_ptr() + gi * self.H_COUNT * 4, self.partials.data_ptr(), npart.value,
                                        self.grad_norm.data_ptr(),
                                        0 if self.flat_p_bf16 is None else self.flat_p_bf16.data_ptr() + start * 2, st)
            _lib.check(rc, "pcm_adamw_flat_hip")

    def step(self):
        self.prepare_step()
        self.launch_step()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step_count": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step_count"])
