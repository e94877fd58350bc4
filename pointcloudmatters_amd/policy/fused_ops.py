"""Fused transformer tail ops backed by csrc/drln.hip, and the per-step context they need.

``drln(x, y, norm, dropout)`` = ``norm(x + dropout(y))`` in one kernel each way.  The dropout mask is a
counter-based hash of (seed, call site, element); the seed lives in a device tensor owned by the
training loop (so hipGraph replays draw fresh masks) and the call-site index is a Python counter that
restarts with every forward -- both come from the active :class:`FusedContext`.  Without an active
context (CPU runs, reference-order tests, eager mode) callers fall back to the framework ops.
"""
import contextlib

import os

import torch
from torch.autograd import Function

from .. import _lib
from . import deferred
from .rows_linear import goes_to_optimizer as _goes_to_optimizer
from .._lib import raw_stream as _raw_stream

_ACTIVE = None


class FusedContext:
    def __init__(self, device, base_seed=0x5EED):
        self.device = torch.device(device)
        self.seed = torch.zeros(1, dtype=torch.int64, device=self.device)
        from .._host import PinnedRing

        self._ring = PinnedRing((1,), torch.int64, self.device)  # a single pinned word rewritten every step would race its DMA
        self.base_seed = int(base_seed)
        self.site = 0
        # set by a training loop that runs backward itself and calls flush_sinks() before it reads the gradients:
        # gradients of a position embedding shared by many attention sites are then summed once (GradSink)
        self.defer_pos_grads = False
        self.sinks = []
        self.counters = []  # BatchNorm num_batches_tracked tensors of this forward pass: incremented together on exit

    def flush_counters(self):
        cs, self.counters = self.counters, []
        if not cs:
            return
        import ctypes

        seen, uniq = set(), []
        for c in cs:  # a module called twice must be counted twice: fall back to single increments for repeats
            if c.data_ptr() in seen:
                c.add_(1)
            else:
                seen.add(c.data_ptr())
                uniq.append(c)
        with torch.cuda.device(self.device):
            rc = _lib.load().pcm_incr_i64_batch_hip(len(uniq), (ctypes.c_void_p * len(uniq))(*[c.data_ptr() for c in uniq]), _raw_stream())
        _lib.check(rc, "pcm_incr_i64_batch_hip")

    def flush_sinks(self):
        for sink in self.sinks:
            sink.flush()

    def set_step(self, step):
        """Host side, between steps / graph replays: an asynchronous 8-byte copy on the current stream."""
        self._ring.next()[0] = self.base_seed + 1000003 * int(step)
        self._ring.push(self.seed)

    def next_site(self):
        self.site += 1
        return self.site


@contextlib.contextmanager
def activate(ctx):
    global _ACTIVE
    prev, _ACTIVE = _ACTIVE, ctx
    if ctx is not None:
        ctx.site = 0
        ctx.sinks = []
    try:
        yield ctx
    finally:
        _ACTIVE = prev
        if ctx is not None and ctx.counters:
            with torch.no_grad():
                ctx.flush_counters()


def current():
    return _ACTIVE


def count_batch(bn):
    """bn.num_batches_tracked += 1: queued on the active FusedContext (one launch for all layers when it exits), else now."""
    ctx, c = _ACTIVE, bn.num_batches_tracked
    if ctx is not None and c is not None and c.is_cuda and c.dtype == torch.int64 and c.device == ctx.device and c.numel() == 1:
        ctx.counters.append(c)
        return
    if c is not None:
        with torch.no_grad():
            c.add_(1)


class GradSink:
    """The decoder adds the SAME query position embedding at 2 sites per layer (transformer.py:330-346): 14 gradients of
    shape (B, L, E) that the autograd engine would add pairwise (13 launches) before the expand-backward sum.  With a
    sink the sites see a detached copy of the embedding, push their gradient here, and ``flush`` adds them in one
    stack + sum and back-propagates the total into the embedding's own graph.  Only used when a FusedContext with
    ``defer_pos_grads`` is active, i.e. by a loop that promises to flush before reading gradients; any consumer that
    cannot push (a non-fused fallback) raises instead of dropping the gradient (transformer._add_pos)."""

    def __init__(self, source):
        self.source = source
        self.grads = []

    def add(self, g):
        self.grads.append(g)

    def flush(self):
        if not self.grads:
            return
        gs, self.grads = self.grads, []
        total = gs[0] if len(gs) == 1 else torch.stack(gs).sum(dim=0)
        torch.autograd.backward([self.source], [total.view_as(self.source)])


def defer_grads(pos):
    """pos (requires grad) -> a detached alias carrying a GradSink, or pos itself when deferral is off."""
    ctx = _ACTIVE
    if ctx is None or not ctx.defer_pos_grads or not pos.requires_grad or not torch.is_grad_enabled():
        return pos
    if not (pos.is_cuda and pos.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return pos  # the consumers that can push (_SelfAttnInProj, _AddPosLinear) are the bf16-autocast nodes
    alias = pos.detach()
    alias._pcm_sink = GradSink(pos)
    ctx.sinks.append(alias._pcm_sink)
    return alias


class _DRLN(Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps, p_drop, seed, site):
        shape = x.shape
        E = shape[-1]
        x2 = x.reshape(-1, E)
        y2 = y.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if not y2.is_contiguous():
            y2 = y2.contiguous()
        out, s, mean, rstd = _drln_forward(x2, y2, gamma, beta, eps, p_drop, seed, site)
        ctx.save_for_backward(s, mean, rstd, gamma)
        ctx.meta = (shape, y.dtype, float(p_drop), seed, int(site))
        ctx.defer = deferred.targets(gamma, beta)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout):
        s, mean, rstd, gamma = ctx.saved_tensors
        shape, ydtype, p_drop, seed, site = ctx.meta
        dx, dy, sums = _drln_backward(dout, s, mean, rstd, gamma, ydtype, p_drop, seed, site, defer=deferred.clear(*ctx.defer))
        return dx.view(shape), dy.view(shape), sums[0], sums[1], None, None, None, None


def _drln_forward(x2, y2, gamma, beta, eps, p_drop, seed, site, emit=None):
    L = _lib.load()
    R, E = x2.shape
    dev = x2.device
    if emit is not None:
        with torch.cuda.device(dev):
            s = torch.empty_like(x2)
            out = deferred.take(x2.shape, x2.dtype, dev, "drln.out")
            mean = torch.empty(R, dtype=torch.float32, device=dev)
            rstd = torch.empty(R, dtype=torch.float32, device=dev)
            extra, rec = _emit_args(emit, R, E, dev)
            rc = L.pcm_drln_forward2_hip(R, E, 1 if y2.dtype == torch.bfloat16 else 0, x2.data_ptr(), y2.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), float(eps), float(p_drop), seed.data_ptr() if seed is not None else 0,
                                         int(site), s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), *extra,
                                         _raw_stream())
        _lib.check(rc, "pcm_drln_forward2_hip")
        emit["record"] = rec
        return out, s, mean, rstd
    with torch.cuda.device(dev):
        s = torch.empty_like(x2)
        out = deferred.take(x2.shape, x2.dtype, dev, "drln.out")
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        rc = L.pcm_drln_forward_hip(R, E, 1 if y2.dtype == torch.bfloat16 else 0, x2.data_ptr(), y2.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), float(eps), float(p_drop), seed.data_ptr() if seed is not None else 0,
                                    int(site), s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                    _raw_stream())
    _lib.check(rc, "pcm_drln_forward_hip")
    return out, s, mean, rstd


# csrc/proj_ln.hip: the output projection, the residual add and the norm as ONE matrix-core kernel (forward; the backward stays
# pcm_drln_backward2_hip + the two products).  Written in round 5 while the GPU pool was closed to the build: verified on the host wave64
# model (tests/test_proj_ln_gpu.py through tests/test_wavesim_parity.py), NOT yet timed -- opt-in until it is (PCM_PROJ_MFMA=1);
# PCM_PROJ_MFMA_MAX_ROWS bounds the row count it takes (a 16-row tile per workgroup streams the whole weight: meant for the ~800-row
# decoder / CVAE-encoder sites, the 4120-row encoder sites stay with the library product unless the bound is raised).
PROJ_MFMA = os.environ.get("PCM_PROJ_MFMA", "0") != "0"
PROJ_MFMA_MAX_ROWS = int(os.environ.get("PCM_PROJ_MFMA_MAX_ROWS", "1024"))
# the LONG sites as well (the encoder's 4120 / 8216 / 16408 rows: from 2048 rows on the kernel takes 64-row tiles, every weight operand
# feeding four matrix instructions); a separate switch because short and long sites can win or lose independently
PROJ_MFMA_LONG = os.environ.get("PCM_PROJ_MFMA_LONG", "0") != "0"
PROJ_MFMA_LONG_ROWS = 2048  # csrc/proj_ln.hip kLongRows


# the BACKWARD of the chain at the short sites (round 6: csrc/proj_ln.hip pcm_proj_drln_mfma_backward = pcm_drln_bwd's row code + da = dy W in one
# launch; host-model-verified, never timed: opt-in like the forward)
PROJ_MFMA_BWD = os.environ.get("PCM_PROJ_MFMA_BWD", "0") != "0"


# ... and the input gradient of the short in-projections / the query projection (pcm_linear_mfma_backward: dx = dy W + dres and the position
# embedding's share in one launch instead of a batched product + an add kernel); same status
LINEAR_MFMA_BWD = os.environ.get("PCM_LINEAR_MFMA_BWD", "0") != "0"


def _linear_bwd_mfma_ok(dy2, ld, N, wc, pos_cols):
    rows = dy2.shape[0]
    return (LINEAR_MFMA_BWD and dy2.is_cuda and 0 < rows <= PROJ_MFMA_MAX_ROWS and dy2.dtype == torch.bfloat16 and wc.dtype == torch.bfloat16
            and wc.is_contiguous() and wc.shape[0] == N and ld % 8 == 0 and dy2.data_ptr() % 16 == 0 and wc.data_ptr() % 16 == 0
            and bool(_lib.load().pcm_linear_mfma_backward_supported(int(N), int(wc.shape[1]), int(pos_cols))))


def _linear_mfma_backward(dy_ptr_tensor, rows, ld, N, wc, dres, want_dpos, pos_cols):
    """dx (rows, K) fp32 = dy (rows, N; row stride ld) @ wc (N, K) [+ dres]; dpos = dy[:, :pos_cols] @ wc[:pos_cols] when wanted."""
    K = int(wc.shape[1])
    dev = dy_ptr_tensor.device
    with torch.cuda.device(dev):
        dx = torch.empty(rows, K, dtype=torch.float32, device=dev)
        dpos = torch.empty(rows, K, dtype=torch.float32, device=dev) if want_dpos else None
        rc = _lib.load().pcm_linear_mfma_backward_hip(rows, int(N), K, dy_ptr_tensor.data_ptr(), int(ld), wc.data_ptr(),
                                                      dres.data_ptr() if dres is not None else 0, dx.data_ptr(),
                                                      dpos.data_ptr() if dpos is not None else 0, int(pos_cols), _raw_stream())
    _lib.check(rc, "pcm_linear_mfma_backward_hip")
    return dx, dpos


LINEAR_MFMA = os.environ.get("PCM_LINEAR_MFMA", "0") != "0"  # csrc/proj_ln.hip pcm_linear_mfma: same status as PROJ_MFMA (opt-in, untimed)


def _linear_mfma(rows, wc, bc, out, *, a16=None, a16_alt=None, x32=None, posc=None, pos_cols=0, emit_pos16=None, emit_x16=None):
    """out (rows, N) = A @ wc^T + bc through csrc/proj_ln.hip's pcm_linear_mfma: A = bf16 operands (a16 for the output columns below
    pos_cols, a16_alt for the others) or fp32 x32 (+ posc below pos_cols), whose bf16 forms go to emit_* as well."""
    N, K = wc.shape
    a = a16 if a16 is not None else x32
    ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(out.device):
        rc = _lib.load().pcm_linear_mfma_forward_hip(rows, int(N), int(K), a.data_ptr(), int(a16 is None), int(a.stride(0)), ptr(a16_alt),
                                                     ptr(posc), posc.numel() if posc is not None else 0, int(pos_cols), wc.data_ptr(),
                                                     ptr(bc), int(bc is not None and bc.dtype == torch.bfloat16), out.data_ptr(),
                                                     int(out.dtype == torch.bfloat16), int(out.stride(0)), ptr(emit_pos16), ptr(emit_x16),
                                                     _raw_stream())
    _lib.check(rc, "pcm_linear_mfma_forward_hip")
    return out


def _linear_mfma_ok(rows, wc, bc, pos_cols, *operands):
    N, K = wc.shape
    # long sites: only the in-projections (N = 3E) and narrower products; the decoder's stacked memory projection (4120 x 3584 x 512 at C2)
    # is a large GEMM that the library's LDS-tiled kernels serve well
    return (LINEAR_MFMA and (0 < rows <= PROJ_MFMA_MAX_ROWS or (PROJ_MFMA_LONG and rows >= PROJ_MFMA_LONG_ROWS and N <= 2048)) and wc.is_cuda and wc.dtype == torch.bfloat16 and wc.is_contiguous()
            and (bc is None or (bc.is_contiguous() and bc.dtype in (torch.bfloat16, torch.float32)))
            and all(t is None or (t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0) for t in operands)
            and wc.data_ptr() % 16 == 0 and bool(_lib.load().pcm_linear_mfma_supported(int(N), int(K), int(pos_cols))))


def _proj_mfma_ok(a2, wc, bc, x2):
    R, E = x2.shape
    return (PROJ_MFMA and a2.is_cuda and (0 < R <= PROJ_MFMA_MAX_ROWS or (PROJ_MFMA_LONG and R >= PROJ_MFMA_LONG_ROWS))
            and a2.dtype == torch.bfloat16 and wc.dtype == torch.bfloat16
            and bc.dtype in (torch.bfloat16, torch.float32) and x2.dtype == torch.float32 and wc.is_contiguous() and bc.is_contiguous()
            and a2.stride(-1) == 1 and a2.stride(0) % 8 == 0 and a2.data_ptr() % 16 == 0 and wc.data_ptr() % 16 == 0
            and bool(_lib.load().pcm_proj_drln_mfma_supported(int(E), int(wc.shape[1]))))


def _proj_drln_forward(a2, wc, bc, x2, gamma, beta, eps, p_drop, seed, site, emit=None):
    """out, s, mean, rstd of LayerNorm(x2 + dropout(a2 @ wc^T + bc)) from one launch (csrc/proj_ln.hip)."""
    L = _lib.load()
    R, E = x2.shape
    dev = x2.device
    with torch.cuda.device(dev):
        s = torch.empty_like(x2)
        out = deferred.take(x2.shape, x2.dtype, dev, "drln.out")
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        extra, rec = _emit_args(emit, R, E, dev)
        rc = L.pcm_proj_drln_mfma_forward_hip(R, E, int(wc.shape[1]), a2.data_ptr(), int(a2.stride(0)), wc.data_ptr(), bc.data_ptr(),
                                              int(bc.dtype == torch.bfloat16), x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                              float(eps), float(p_drop), seed.data_ptr() if seed is not None else 0, int(site),
                                              s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), *extra, _raw_stream())
    _lib.check(rc, "pcm_proj_drln_mfma_forward_hip")
    if emit is not None:
        emit["record"] = rec
    return out, s, mean, rstd


def _pos_rows(pos, x_shape):
    """`pos` as the contiguous fp32 block that the add + cast kernels broadcast over the leading rows of x (shape x_shape):
    trailing dimensions expanded to x's, a batch-broadcast view reduced to its single row block."""
    nd = len(x_shape)
    posc = pos.expand(pos.shape[0], *x_shape[1:]) if pos.dim() == nd and tuple(pos.shape[1:]) != tuple(x_shape[1:]) else pos
    if posc.dim() == nd and posc.shape[0] > 1 and posc.stride(0) == 0:
        posc = posc[:1]  # a batch-broadcast view (query_pos): the kernel broadcasts by index, no materialised copy
    return posc.contiguous()


def _emit_args(emit, R, E, dev):
    """emit = {"pos": tensor, "x16": bool, "tags": (tag of bf16(out + pos), tag of bf16(out))} -> the extra arguments of
    pcm_*_forward2_hip and the record that the consumer node looks for on its input (``_emitted``)."""
    if emit is None:
        return (0, 0, 0, 0), None
    posc = emit["posc"]
    if emit["x16"]:  # the in-projection's pair: one (2, R, E) buffer, so that it can run as ONE product ([x + pos ; x])
        sum16, x16 = deferred.take((2, R, E), torch.bfloat16, dev, emit["tags"][0]).unbind(0)
    else:
        sum16, x16 = deferred.take((R, E), torch.bfloat16, dev, emit["tags"][0]), None
    return (posc.data_ptr(), posc.numel(), sum16.data_ptr(), x16.data_ptr() if x16 is not None else 0), \
        {"pos": emit["pos"], "posc": posc, "sum16": sum16, "x16": x16}


EMIT_OPERANDS = os.environ.get("PCM_EMIT_OPERANDS", "1") != "0"  # A/B switch (and tests: bit-identical either way)
INPROJ_MERGE_ROWS = int(os.environ.get("PCM_INPROJ_MERGE_ROWS", 2048))  # below this many rows q | k and v come from ONE product


def emit_for(pos, x_shape, x16, tags):
    """What a producer node (proj_drln / ffn_ln) needs to write its consumer's bf16 operands in its own launch: the consumer
    (an in-projection or query-projection node fed with THIS pos object) then skips its add + cast kernel.  None when the
    shapes do not allow it."""
    if not EMIT_OPERANDS or pos is None or not pos.is_cuda or pos.dtype != torch.float32 or not (
            torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16) or not torch.is_grad_enabled():
        return None
    E = x_shape[-1]
    numel = 1
    for v in x_shape:
        numel *= int(v)
    if pos.dim() != len(x_shape) or tuple(pos.shape[1:]) != tuple(x_shape[1:]) or pos.shape[0] not in (1, x_shape[0]):
        return None
    posc = _pos_rows(pos, x_shape)
    if posc.numel() % E or numel % posc.numel():
        return None
    return {"pos": pos, "posc": posc, "x16": bool(x16), "tags": tags}


def _emitted(x, pos, posc):
    """The bf16 operands a producer already wrote for this (x, pos) pair, or None."""
    em = getattr(x, "_pcm_emit", None)
    if em is None or em["pos"] is not pos or em["posc"].shape != posc.shape or em["sum16"].numel() != x.numel():
        return None
    return em


def _f32_rows(g, R, E):
    g = g.reshape(R, E)
    return g if g.dtype == torch.float32 and g.is_contiguous() else g.float().contiguous()


def _two_addends(douts, R, E):
    """The gradients of a node's aliased outputs -> (dout, dout2 or None): two are summed inside the backward kernel."""
    gs = [_f32_rows(g, R, E) for g in douts if g is not None]
    if not gs:
        return None, None
    while len(gs) > 2:
        gs = [gs[0] + gs[1]] + gs[2:]
    return gs[0], (gs[1] if len(gs) > 1 else None)


def _drln_backward(dout, s, mean, rstd, gamma, ydtype, p_drop, seed, site, dysum_bf16=False, defer=False, dout2=None):
    """-> dx (R,E) fp32, dy (R,E) in ydtype, sums (3,E) fp32 = dgamma | dbeta | column sums of dy (+ those sums in bf16).
    defer: leave the closing reduction of `sums` to policy/deferred.flush (the caller checked deferred.clear)."""
    L = _lib.load()
    R, E = s.shape
    dev = s.device
    d2 = dout.reshape(R, E)
    if d2.dtype != torch.float32 or not d2.is_contiguous():
        d2 = d2.float().contiguous()
    with torch.cuda.device(dev):
        dx = torch.empty_like(s)
        dy = deferred.take((R, E), ydtype, dev, "drln.dy")
        blocks = L.pcm_drln_blocks(R)
        partial = torch.empty(blocks * 3 * E, dtype=torch.float32, device=dev)
        sums = torch.empty(3, E, dtype=torch.float32, device=dev)
        db16 = torch.empty(E, dtype=torch.bfloat16, device=dev) if dysum_bf16 else None
        defer = defer and R > 0 and deferred.push(partial, blocks, 3 * E, out_f32=sums, out_bf16=db16, bf16_from=2 * E)
        rc = L.pcm_drln_backward2_hip(R, E, 1 if ydtype == torch.bfloat16 else 0, d2.data_ptr(),
                                      dout2.data_ptr() if dout2 is not None else 0, s.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), gamma.data_ptr(), p_drop, seed.data_ptr() if seed is not None else 0, site,
                                      dx.data_ptr(), dy.data_ptr(), partial.data_ptr(), 0 if defer else sums.data_ptr(),
                                      db16.data_ptr() if db16 is not None else 0, _raw_stream())
    _lib.check(rc, "pcm_drln_backward2_hip")
    if dysum_bf16:
        return dx, dy, sums, (deferred.handout(db16) if defer else db16)
    return dx, dy, sums


def _proj_bwd_mfma_ok(s, wc, ydt, adt):
    R, E = s.shape
    return (PROJ_MFMA_BWD and s.is_cuda and 0 < R <= PROJ_MFMA_MAX_ROWS and ydt == torch.bfloat16 and adt == torch.bfloat16
            and wc.dtype == torch.bfloat16 and wc.is_contiguous() and wc.shape[0] == E and wc.data_ptr() % 16 == 0
            and bool(_lib.load().pcm_proj_drln_mfma_backward_supported(int(E), int(wc.shape[1]))))


def _proj_drln_backward_mfma(dout, s, mean, rstd, gamma, wc, p_drop, seed, site, dysum_bf16=False, defer=False, dout2=None):
    """`_drln_backward` and the projection's input gradient from ONE launch (csrc/proj_ln.hip):
    -> dx (R,E) fp32, dy (R,E) bf16, sums (3,E) fp32 [, column sums of dy in bf16], da (R,K) bf16 = dy @ wc."""
    L = _lib.load()
    R, E = s.shape
    K = int(wc.shape[1])
    dev = s.device
    d2 = dout.reshape(R, E)
    if d2.dtype != torch.float32 or not d2.is_contiguous():
        d2 = d2.float().contiguous()
    with torch.cuda.device(dev):
        dx = torch.empty_like(s)
        dy = deferred.take((R, E), torch.bfloat16, dev, "drln.dy")
        da = torch.empty((R, K), dtype=torch.bfloat16, device=dev)
        blocks = L.pcm_proj_drln_mfma_backward_blocks(R)
        partial = torch.empty(blocks * 3 * E, dtype=torch.float32, device=dev)
        sums = torch.empty(3, E, dtype=torch.float32, device=dev)
        db16 = torch.empty(E, dtype=torch.bfloat16, device=dev) if dysum_bf16 else None
        defer = defer and deferred.push(partial, blocks, 3 * E, out_f32=sums, out_bf16=db16, bf16_from=2 * E)
        rc = L.pcm_proj_drln_mfma_backward_hip(R, E, K, d2.data_ptr(), dout2.data_ptr() if dout2 is not None else 0, s.data_ptr(),
                                               mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), p_drop,
                                               seed.data_ptr() if seed is not None else 0, site, wc.data_ptr(), dx.data_ptr(), dy.data_ptr(),
                                               da.data_ptr(), int(da.stride(0)), partial.data_ptr(), 0 if defer else sums.data_ptr(),
                                               db16.data_ptr() if db16 is not None else 0, _raw_stream())
    _lib.check(rc, "pcm_proj_drln_mfma_backward_hip")
    if dysum_bf16:
        return dx, dy, sums, (deferred.handout(db16) if defer else db16), da
    return dx, dy, sums, da


class _ProjDRLN(Function):
    """out = LayerNorm(x + dropout(a @ W^T + b)): the attention output projection, the residual add and the norm as one
    autograd node.  Forward = one GEMM + the drln kernel; backward = the drln kernel (which also yields the column
    sums of dy = the projection's bias gradient, saving the framework's strided reduction), one GEMM for da and a
    split-K product for dW (rows_linear.weight_grad)."""

    @staticmethod
    def forward(ctx, a, weight, bias, x, gamma, beta, eps, p_drop, seed, site, n_out=1, emit=None):
        shape = x.shape
        E = shape[-1]
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            ac, wc, bc = a.to(dt), weight.to(dt), bias.to(dt)
        else:
            ac, wc, bc = a, weight, bias
        a2 = ac.reshape(-1, ac.shape[-1])
        x2 = x.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if _proj_mfma_ok(a2, wc, bc, x2):
            ydt = torch.bfloat16  # the kernel rounds the product to bf16, like the library GEMM under autocast
            out, s, mean, rstd = _proj_drln_forward(a2, wc, bc, x2, gamma, beta, eps, p_drop, seed, site, emit=emit)
        else:
            with torch.autocast("cuda", enabled=False):
                y2 = torch.nn.functional.linear(a2, wc, bc)
            ydt = y2.dtype
            out, s, mean, rstd = _drln_forward(x2, y2, gamma, beta, eps, p_drop, seed, site, emit=emit)
        ctx.save_for_backward(a2, wc, s, mean, rstd, gamma)
        ctx.meta = (shape, a.shape, a.dtype, weight.dtype, bias.dtype, ydt, float(p_drop), seed, int(site))
        ctx.side_ok = _goes_to_optimizer(weight)
        ok, leaves = deferred.targets(weight, bias, gamma, beta)
        ctx.defer = (ok and bias.dtype in (torch.bfloat16, torch.float32), leaves)
        if n_out == 1:
            return out.view(shape)
        # one alias per consumer: the engine hands their gradients over separately and the backward kernel sums them while
        # loading, instead of an add launch in between
        ctx.set_materialize_grads(False)
        return tuple(out.view(shape) for _ in range(n_out))

    @staticmethod
    def backward(ctx, *douts):
        from .rows_linear import weight_grad

        a2, wc, s, mean, rstd, gamma = ctx.saved_tensors
        shape, ashape, adt, wdt, bdt, ydt, p_drop, seed, site = ctx.meta
        want16 = bdt == torch.bfloat16
        dout, dout2 = _two_addends(douts, *s.shape)
        if dout is None:
            return (None,) * 12
        defer = deferred.clear(*ctx.defer)
        fused_da = _proj_bwd_mfma_ok(s, wc, ydt, a2.dtype)
        if fused_da:  # the row code and da = dy W in one launch (opt-in, PROJ_MFMA_BWD)
            res = _proj_drln_backward_mfma(dout, s, mean, rstd, gamma, wc, p_drop, seed, site, dysum_bf16=want16, defer=defer, dout2=dout2)
        else:
            res = _drln_backward(dout, s, mean, rstd, gamma, ydt, p_drop, seed, site, dysum_bf16=want16, defer=defer, dout2=dout2)
        dx, dy, sums = res[:3]
        with torch.autocast("cuda", enabled=False):
            da = (res[-1] if fused_da else dy @ wc).view(ashape)
            if da.dtype != adt:
                da = da.to(adt)
            dw = weight_grad(dy, a2, wdt, side=ctx.side_ok, defer=defer, tag="proj_drln")
            db = res[3] if want16 else sums[2].to(bdt)
        return da, dw, db, dx.view(shape), sums[0], sums[1], None, None, None, None, None, None


def _tag_emitted(outs, emit):
    rec = emit.get("record") if emit is not None else None
    if rec is not None:
        for o in (outs if isinstance(outs, tuple) else (outs,)):
            o._pcm_emit = rec
    return outs


def proj_drln(a, linear, x, norm, dropout, n_out=1, emit=None):
    """norm(x + dropout(linear(a))); the caller checked ``drln_supported(x, <linear output>, norm)``.  n_out > 1: that many
    aliases of the result, one per consumer (their gradients are summed inside the backward kernel).  emit (``emit_for``):
    the same launch also writes the bf16 operands of the node that will consume the result together with that pos."""
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    ctx = _ACTIVE
    return _tag_emitted(_ProjDRLN.apply(a, linear.weight, linear.bias, x, norm.weight, norm.bias, norm.eps, p,
                                        ctx.seed if p > 0 else None, ctx.next_site(), n_out, emit), emit)


def drln_supported(x, y, norm, y_dtype=None):
    """`y` may be None when only its dtype is known yet (the projection that produces it is part of the fused node)."""
    e = x.shape[-1]
    ydt = y.dtype if y is not None else y_dtype
    return (_ACTIVE is not None and x.is_cuda and x.dtype == torch.float32 and ydt in (torch.float32, torch.bfloat16)
            and type(norm) is torch.nn.LayerNorm and norm.elementwise_affine and norm.bias is not None
            and e % 256 == 0 and e <= 1024 and (y is None or x.shape == y.shape))


def drln(x, y, norm, dropout):
    """norm(x + dropout(y)) through the fused kernel; the caller checked ``drln_supported``."""
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    ctx = _ACTIVE
    return _DRLN.apply(x, y, norm.weight, norm.bias, norm.eps, p, ctx.seed if p > 0 else None, ctx.next_site())


# csrc/ffn_mfma.hip (the sub-layer's two products on the matrix cores, bf16-autocast roundings) is OPT-IN: measured on MI355X by graph
# replay (tools/mb/mb_ffn.py) it is 18.0 / 18.1 us forward / backward at 4120 rows against 20.5 / 20.1 us for the fp32 kernel, and
# 10.4 / 9.2 us against 8.8 / 9.2 us at the decoder's 800 rows -- a 16-row tile is a chain of dependent latencies (operand loads,
# three LDS exchanges with barriers) with one wave per SIMD, ~26 k clocks whatever the row count, not 0.3 GFLOP
# of arithmetic.  PCM_FFN_MFMA=1 enables it from `FFN_MFMA_MIN_ROWS` rows on.
FFN_MFMA = os.environ.get("PCM_FFN_MFMA", "0") != "0"
FFN_MFMA_MIN_ROWS = int(os.environ.get("PCM_FFN_MFMA_MIN_ROWS", "2048"))


class _FFNLN(Function):
    """out = norm(x + dropout_out(linear2(dropout_hidden(relu(linear1(x)))))) -- csrc/ffn.hip."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, p_hidden, p_out, seed, site_a, site_b, n_out=1, emit=None):
        L = _lib.load()
        shape = x.shape
        E, Fh = shape[-1], w1.shape[0]
        x2 = x.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        R = x2.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hd, s, out = deferred.take((R, Fh), torch.float32, dev, "ffn.hd"), torch.empty(R, E, **f32), deferred.take((R, E), torch.float32, dev, "drln.out")
            mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
            sp = seed.data_ptr() if seed is not None else 0
            extra, rec = _emit_args(emit, R, E, dev)
            # bf16 autocast: the two products are bf16 GEMMs in the reference recipe -> the matrix-core kernel (csrc/ffn_mfma.hip);
            # fp32 runs keep the fp32 kernel (csrc/ffn.hip).  Opt-in, see FFN_MFMA above
            mfma = FFN_MFMA and R >= FFN_MFMA_MIN_ROWS and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 \
                and bool(L.pcm_ffn_ln_mfma_supported(E, Fh))
            fwd = L.pcm_ffn_ln_mfma_forward_hip if mfma else L.pcm_ffn_ln_forward2_hip
            rc = fwd(R, E, Fh, x2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(), float(eps), float(p_hidden), float(p_out), sp,
                                           int(site_a), int(site_b), hd.data_ptr(), s.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), *extra, _raw_stream())
            if emit is not None:
                emit["record"] = rec
        _lib.check(rc, "pcm_ffn_ln_mfma_forward_hip" if mfma else "pcm_ffn_ln_forward2_hip")
        ctx.save_for_backward(x2, w1, w2, gamma, hd, s, mean, rstd)
        ctx.meta = (shape, float(p_hidden), float(p_out), seed, int(site_b), mfma)
        ctx.side_ok = _goes_to_optimizer(w1) and _goes_to_optimizer(w2)
        ctx.defer = deferred.targets(w1, b1, w2, b2, gamma, beta)
        if n_out == 1:
            return out.view(shape)
        ctx.set_materialize_grads(False)  # one alias per consumer, see _ProjDRLN.forward
        return tuple(out.view(shape) for _ in range(n_out))

    @staticmethod
    def backward(ctx, *douts):
        L = _lib.load()
        x2, w1, w2, gamma, hd, s, mean, rstd = ctx.saved_tensors
        shape, p_hidden, p_out, seed, site_b, mfma = ctx.meta
        R, E = x2.shape
        Fh = w1.shape[0]
        dev = x2.device
        f32 = dict(dtype=torch.float32, device=dev)
        d2, d2b = _two_addends(douts, R, E)
        if d2 is None:
            return (None,) * 15
        with torch.cuda.device(dev):
            dx, dy, dh = torch.empty(R, E, **f32), deferred.take((R, E), torch.float32, dev, "ffn.dy"), deferred.take((R, Fh), torch.float32, dev, "ffn.dh")
            pw = 3 * E + Fh
            blocks = L.pcm_ffn_ln_mfma_blocks(R) if mfma else L.pcm_ffn_ln_blocks(R)
            partial = torch.empty(blocks * pw, **f32)
            sums = torch.empty(pw, **f32)
            defer = deferred.clear(*ctx.defer)
            defer_sums = defer and R > 0 and deferred.push(partial, blocks, pw, out_f32=sums)
            bwd = L.pcm_ffn_ln_mfma_backward_hip if mfma else L.pcm_ffn_ln_backward2_hip
            rc = bwd(R, E, Fh, d2.data_ptr(), d2b.data_ptr() if d2b is not None else 0, x2.data_ptr(),
                                            s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hd.data_ptr(), w1.data_ptr(),
                                            w2.data_ptr(), gamma.data_ptr(), p_hidden, p_out,
                                            seed.data_ptr() if seed is not None else 0, site_b, dx.data_ptr(), dy.data_ptr(),
                                            dh.data_ptr(), partial.data_ptr(), 0 if defer_sums else sums.data_ptr(),
                                            _raw_stream())
            _lib.check(rc, "pcm_ffn_ln_mfma_backward_hip" if mfma else "pcm_ffn_ln_backward2_hip")
            from .rows_linear import weight_grad

            with torch.autocast(device_type="cuda", enabled=False):
                dw2 = weight_grad(dy, hd, torch.float32, side=ctx.side_ok, defer=defer, tag="ffn.w2")  # (E, F)   split-K over the rows when there are thousands
                dw1 = weight_grad(dh, x2, torch.float32, side=ctx.side_ok, defer=defer, tag="ffn.w1")  # (F, E)
        dgamma, dbeta, db2, db1 = sums[:E], sums[E : 2 * E], sums[2 * E : 3 * E], sums[3 * E :]
        return dx.view(shape), dw1, db1, dw2, db2, dgamma, dbeta, None, None, None, None, None, None, None, None


def ffn_ln_supported(x, linear1, linear2, norm):
    if _ACTIVE is None or not x.is_cuda or x.dtype != torch.float32:
        return False
    ws = (linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight, norm.bias)
    if any(w is None or w.dtype != torch.float32 for w in ws) or type(norm) is not torch.nn.LayerNorm:
        return False
    return bool(_lib.load().pcm_ffn_ln_supported(int(x.shape[-1]), int(linear1.weight.shape[0])))


def ffn_ln(x, linear1, linear2, norm, dropout_hidden, dropout_out, n_out=1, emit=None):
    pa = dropout_hidden.p if dropout_hidden.training else 0.0
    pb = dropout_out.p if dropout_out.training else 0.0
    ctx = _ACTIVE
    return _tag_emitted(_FFNLN.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight, norm.bias, norm.eps,
                                     pa, pb, ctx.seed if (pa > 0 or pb > 0) else None, ctx.next_site(), ctx.next_site(), n_out,
                                     emit), emit)


class _SelfAttnInProj(Function):
    """q, k = split(linear(x + pos, W[:2E], b[:2E])), v = linear(x, W[2E:], b[2E:]) for a packed in_proj (3E, E) under
    bf16 autocast, as ONE autograd node: csrc/tokens.hip does the add + casts (1 launch forward, 1 backward) and the
    three bias gradients (2 launches); weight gradients are split-K products written straight into the packed (3E, E)
    gradient, so no cat / split-backward kernels run."""

    @staticmethod
    def forward(ctx, x, pos, w, b):
        L = _lib.load()
        shape = x.shape
        E = shape[-1]
        x2 = x.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        posc = _pos_rows(pos, shape)
        dev = x.device
        bf = torch.bfloat16
        wc = w if w.dtype == bf else w.to(bf)
        bc = b if b.dtype == bf else b.to(bf)
        em = _emitted(x, pos, posc)
        have16 = em is not None and em["x16"] is not None  # the producer of x wrote bf16(x + pos) and bf16(x) in its own launch
        mfma = _linear_mfma_ok(rows, wc, bc, 2 * E, em["sum16"], em["x16"]) if have16 else _linear_mfma_ok(rows, wc, bc, 2 * E, x2)
        if have16:
            qk_in, v_in = em["sum16"], em["x16"]
        else:
            with torch.cuda.device(dev):
                qk_in, v_in = deferred.take((2, rows, E), bf, dev, "in_proj.qkv").unbind(0)  # adjacent: see the merged product
                if not mfma:
                    rc = L.pcm_add_cast2_hip(x2.numel(), posc.numel(), x2.data_ptr(), posc.data_ptr(), qk_in.data_ptr(), v_in.data_ptr(),
                                             _raw_stream())
                    _lib.check(rc, "pcm_add_cast2_hip")
        with torch.autocast("cuda", enabled=False):
            if mfma:
                # ONE launch: the add + casts happen on the way into LDS (or the emitted operands are read), q | k | v in one (rows, 3E)
                # matrix without the doubled-row trick below; the bf16 operands are written out for the weight-gradient products
                with torch.cuda.device(dev):
                    y = torch.empty(rows, 3 * E, dtype=bf, device=dev)
                if have16:
                    _linear_mfma(rows, wc, bc, y, a16=qk_in, a16_alt=v_in, pos_cols=2 * E)
                else:
                    _linear_mfma(rows, wc, bc, y, x32=x2, posc=posc, pos_cols=2 * E, emit_pos16=qk_in, emit_x16=v_in)
                qk = y[:, : 2 * E].unflatten(-1, (2, E)).unflatten(0, shape[:-1])
                v = y[:, 2 * E:].unflatten(0, shape[:-1])
            elif (INPROJ_MERGE_ROWS > rows and v_in.data_ptr() - qk_in.data_ptr() == rows * E * 2 and qk_in.is_contiguous()
                    and v_in.is_contiguous() and qk_in.untyped_storage().data_ptr() == v_in.untyped_storage().data_ptr()):
                # short activations (the decoder's 800 rows): ONE product [x + pos ; x] (2R, E) @ W^T (E, 3E) instead of two -- twice
                # the arithmetic (the off-diagonal blocks are discarded), but these products sit at the launch floor: 11 us vs
                # 10 + 8.  q | k and v are strided views of its result (the attention kernels take row strides).
                y = torch.nn.functional.linear(torch.as_strided(qk_in, (2 * rows, E), (E, 1)), wc, bc)
                qk = y[:rows, : 2 * E].unflatten(-1, (2, E)).unflatten(0, shape[:-1])
                v = y[rows:, 2 * E:].unflatten(0, shape[:-1])
            else:
                qk = torch.nn.functional.linear(qk_in, wc[: 2 * E], bc[: 2 * E]).view(*shape[:-1], 2, E)
                v = torch.nn.functional.linear(v_in, wc[2 * E:], bc[2 * E:]).view(shape)
        ctx.save_for_backward(qk_in, v_in, wc)
        ctx.sink = getattr(pos, "_pcm_sink", None)
        ctx.meta = (shape, pos.shape, w.dtype, b.dtype, pos.requires_grad or ctx.sink is not None)
        ctx.side_ok = _goes_to_optimizer(w)
        ctx.defer = deferred.targets(w, b)
        q, k = qk.unbind(-2)
        # x is handed back as a fourth output for the caller's residual branch: x then has ONE consumer in the autograd
        # graph and this node receives the residual's gradient, which it folds into its closing add kernel (otherwise the
        # engine adds the two gradients of x with a launch of its own)
        return q, k, v, x.view_as(x)

    @staticmethod
    def backward(ctx, dq, dk, dv, dres=None):
        from .rows_linear import weight_grad

        L = _lib.load()
        qk_in, v_in, wc = ctx.saved_tensors
        shape, pos_shape, wdt, bdt, pos_grad = ctx.meta
        E = shape[-1]
        rows = qk_in.shape[0]
        dev = qk_in.device
        bf = torch.bfloat16
        st = _raw_stream()
        with torch.cuda.device(dev), torch.autocast("cuda", enabled=False):
            es2 = 2  # bytes per bf16
            joint = (dq.dtype == bf and dk.dtype == bf and dv.dtype == bf and dq.stride() == dk.stride() == dv.stride()
                     and dq.stride()[-2:] == (3 * E, 1) and dk.data_ptr() - dq.data_ptr() == E * es2
                     and dv.data_ptr() - dq.data_ptr() == 2 * E * es2
                     and all(dq.stride(i) == dq.stride(i + 1) * dq.shape[i + 1] for i in range(dq.dim() - 2)))
            fused_dx = joint and _linear_bwd_mfma_ok(torch.as_strided(dq, (rows, 3 * E), (3 * E, 1)), 3 * E, 3 * E, wc, 2 * E)
            if joint:
                # dq | dk | dv side by side (small_attn's layout for short self-attention): the three input gradients as ONE
                # batched product (3 x (rows, E) @ (E, E): 9 us; dqk @ W_qk and dv @ W_v as two: 14.5 us) -- or, opt-in, as one
                # matrix-core launch that also adds the residual's gradient and splits off the position share (LINEAR_MFMA_BWD)
                if not fused_dx:
                    d3 = torch.bmm(torch.as_strided(dq, (3, rows, E), (E, 3 * E, 1)), wc.view(3, E, E))
                dqk = torch.as_strided(dq, (rows, 2 * E), (3 * E, 1))
                dv2 = torch.as_strided(dv, (rows, E), (3 * E, 1))
                ld_qk = ld_v = 3 * E
            elif (dq.dtype == bf and dk.dtype == bf and dq.stride() == dk.stride() and dq.stride()[-2:] == (2 * E, 1)
                    and dk.data_ptr() - dq.data_ptr() == 2 * E and dq.is_contiguous() is False
                    and all(dq.stride(i) == dq.stride(i + 1) * dq.shape[i + 1] for i in range(dq.dim() - 2))):
                # the small-attention backward already wrote dq | dk side by side: view them as one (rows, 2E) matrix
                dqk = torch.as_strided(dq, (rows, 2 * E), (2 * E, 1))
            else:
                dqk = torch.stack((dq, dk), dim=-2).reshape(rows, 2 * E)
                if dqk.dtype != bf:
                    dqk = dqk.to(bf)
            if not joint:
                dv2 = dv.reshape(rows, E)
                if dv2.dtype != bf or not dv2.is_contiguous():
                    dv2 = dv2.to(bf).contiguous()
                d_qk_in = dqk @ wc[: 2 * E]
                d_v_in = dv2 @ wc[2 * E:]
                ld_qk, ld_v = 2 * E, E
            if dres is not None:
                dres = dres.reshape(rows, E)
                if dres.dtype != torch.float32 or not dres.is_contiguous():
                    dres = dres.float().contiguous()
            dpos_wide = None
            same_shape = pos_grad and tuple(pos_shape) == tuple(shape)
            if fused_dx:  # one launch: dx = [dq dk dv] W_in + dres, the position share [dq dk] W_qk split off at k = 2E
                dx, dpos_wide = _linear_mfma_backward(dq, rows, 3 * E, 3 * E, wc, dres, pos_grad, 2 * E)
                dpos32 = dpos_wide if same_shape else None
            else:
                dx = torch.empty(rows, E, dtype=torch.float32, device=dev)
                # the position gradient is d_qk_in alone: widened by the same launch when it has the shape of x (query_pos)
                dpos32 = torch.empty(rows, E, dtype=torch.float32, device=dev) if same_shape else None
                if joint:
                    rc = L.pcm_add4_cast2_hip(dx.numel(), d3[0].data_ptr(), d3[1].data_ptr(), d3[2].data_ptr(),
                                              dres.data_ptr() if dres is not None else 0, dx.data_ptr(),
                                              dpos32.data_ptr() if dpos32 is not None else 0, st)
                else:
                    rc = L.pcm_add3_cast2_hip(dx.numel(), d_qk_in.data_ptr(), d_v_in.data_ptr(), dres.data_ptr() if dres is not None else 0,
                                              dx.data_ptr(), dpos32.data_ptr() if dpos32 is not None else 0, st)
                _lib.check(rc, "pcm_add3_cast2_hip")
            dw = deferred.take((3 * E, E), wdt, dev, "in_proj.dw")
            defer = deferred.clear(*ctx.defer)
            weight_grad(dqk, qk_in, wdt, out=dw[: 2 * E], side=ctx.side_ok, defer=defer, tag="in_proj.qk")
            weight_grad(dv2, v_in, wdt, out=dw[2 * E:], side=ctx.side_ok, defer=defer, tag="in_proj.v")
            db = torch.empty(3 * E, dtype=bdt, device=dev)
            slots = L.pcm_colsum_slots(rows, E)
            partial = torch.empty(slots * 3 * E, dtype=torch.float32, device=dev)
            es = dqk.element_size()
            defer = defer and deferred.push(partial, slots, 3 * E, **({"out_bf16": db} if bdt == bf else {"out_f32": db}))
            if not (defer and deferred.push_colsum(rows, E, [dqk.data_ptr(), dqk.data_ptr() + E * es, dv2.data_ptr()], [ld_qk, ld_qk, ld_v],
                                                  bf, partial, (dqk, dv2))):
                rc = L.pcm_colsum_hip(rows, E, 3, 1, dqk.data_ptr(), ld_qk, dqk.data_ptr() + E * es, ld_qk, dv2.data_ptr(), ld_v,
                                      partial.data_ptr(), int(bdt == bf), 0 if defer else db.data_ptr(), st)
                _lib.check(rc, "pcm_colsum_hip")
            dpos = None
            if dpos32 is not None:
                dpos = dpos32.view(shape)
            elif pos_grad:
                dpos = (dpos_wide if fused_dx else ((d3[0].float() + d3[1].float()) if joint else d_qk_in.float())).view(shape).sum_to_size(pos_shape)
            if ctx.sink is not None and dpos is not None:
                ctx.sink.add(dpos)
                dpos = None
        # pending results go to autograd as fresh aliases (deferred.handout): the queues hold views of dw / db
        return dx.view(shape), dpos, deferred.handout(dw), deferred.handout(db)


class _AddPosLinear(Function):
    """y = linear(x + pos, w, b) under bf16 autocast as ONE autograd node (the decoder's cross-attention query
    projection): csrc/tokens.hip adds and casts in one launch; backward writes the input gradient in fp32 straight from
    the GEMM (no cast launches either way) and hands the SAME tensor to x and pos (they enter as a sum)."""

    @staticmethod
    def forward(ctx, x, pos, w, b):
        L = _lib.load()
        shape = x.shape
        E = shape[-1]
        x2 = x.reshape(-1, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        posc = _pos_rows(pos, shape)
        bf = torch.bfloat16
        wc = w if w.dtype == bf else w.to(bf)
        bc = b if b.dtype == bf else b.to(bf)
        em = _emitted(x, pos, posc)
        rows, N = x2.shape[0], wc.shape[0]
        mfma = _linear_mfma_ok(rows, wc, bc, N, em["sum16"] if em is not None else x2)
        if em is not None:  # the producer of x wrote bf16(x + pos) in its own launch
            s_in = em["sum16"]
        else:
            with torch.cuda.device(x.device):
                s_in = deferred.take(x2.shape, bf, x.device, "add_pos.s")
                if not mfma:
                    rc = L.pcm_add_cast2_hip(x2.numel(), posc.numel(), x2.data_ptr(), posc.data_ptr(), s_in.data_ptr(), 0,
                                             _raw_stream())
                    _lib.check(rc, "pcm_add_cast2_hip")
        with torch.autocast("cuda", enabled=False):
            if mfma:  # one launch; bf16(x + pos) is still written out: it is the weight-gradient product's operand
                with torch.cuda.device(x.device):
                    y = torch.empty(rows, N, dtype=bf, device=x.device)
                if em is not None:
                    _linear_mfma(rows, wc, bc, y, a16=s_in, pos_cols=N)
                else:
                    _linear_mfma(rows, wc, bc, y, x32=x2, posc=posc, pos_cols=N, emit_pos16=s_in)
                y = y.view(*shape[:-1], N)
            else:
                y = torch.nn.functional.linear(s_in, wc, bc).view(*shape[:-1], wc.shape[0])
        ctx.save_for_backward(s_in, wc)
        ctx.meta = (shape, pos.shape, w.dtype, b.dtype)
        ctx.side_ok = _goes_to_optimizer(w)
        ctx.defer = deferred.targets(w)
        ctx.defer_b = deferred.targets(b)
        ctx.sink = getattr(pos, "_pcm_sink", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .rows_linear import weight_grad

        s_in, wc = ctx.saved_tensors
        shape, pos_shape, wdt, bdt = ctx.meta
        bf = torch.bfloat16
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != bf or not dy2.is_contiguous():
            dy2 = dy2.to(bf).contiguous()
        dx = dpos = dw = db = None
        with torch.cuda.device(dy.device), torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.sink is not None:
                if _linear_bwd_mfma_ok(dy2, int(dy2.stride(0)), int(wc.shape[0]), wc, int(wc.shape[0])):  # opt-in, LINEAR_MFMA_BWD
                    d_in = _linear_mfma_backward(dy2, dy2.shape[0], int(dy2.stride(0)), int(wc.shape[0]), wc, None, False, int(wc.shape[0]))[0].view(shape)
                else:
                    d_in = torch.mm(dy2, wc, out_dtype=torch.float32).view(shape)  # fp32 out of the bf16 GEMM: no cast kernel
                dx = d_in if ctx.needs_input_grad[0] else None
                if ctx.sink is not None:
                    ctx.sink.add(d_in.sum_to_size(pos_shape))
                else:
                    dpos = d_in.sum_to_size(pos_shape) if ctx.needs_input_grad[1] else None
            if ctx.needs_input_grad[2]:
                dw = weight_grad(dy2, s_in, wdt, side=ctx.side_ok, defer=deferred.clear(*ctx.defer), tag="add_pos")
            if ctx.needs_input_grad[3]:
                from .rows_linear import bias_grad

                db = bias_grad(dy2, bdt, defer=deferred.clear(*ctx.defer_b))
        return dx, dpos, dw, db


def add_pos_linear_supported(x, pos, w, b):
    return (_ACTIVE is not None and x.is_cuda and x.dtype == torch.float32 and pos is not None and pos.dtype == torch.float32
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and b is not None
            and x.shape[-1] % 4 == 0 and pos.dim() == x.dim() and pos.shape[1:] == x.shape[1:] and pos.shape[0] in (1, x.shape[0])
            and w.dtype in (torch.float32, torch.bfloat16) and torch.is_grad_enabled())


def add_pos_linear(x, pos, w, b):
    return _AddPosLinear.apply(x, pos, w, b)


def self_attn_in_proj_supported(x, pos, mha):
    e = x.shape[-1]
    return (_ACTIVE is not None and x.is_cuda and x.dtype == torch.float32 and pos is not None and pos.dtype == torch.float32
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and mha.in_proj_weight is not None and mha.in_proj_bias is not None and e % 4 == 0 and e <= 1024
            and mha.in_proj_bias.dtype in (torch.float32, torch.bfloat16) and x.numel() % pos.numel() == 0 and pos.dim() == x.dim()
            and (pos.shape[0] in (1, x.shape[0])) and pos.shape[1:] == x.shape[1:])


def self_attn_in_proj(x, pos, mha):
    """q, k, v (each (B, S, E) bf16) for self-attention with position-augmented queries / keys, and x again (an alias
    for the residual branch, see _SelfAttnInProj.forward)."""
    return _SelfAttnInProj.apply(x, pos, mha.in_proj_weight, mha.in_proj_bias)


# ----------------------------------------------------------------------------------------------------------------------
# ACT training loss in one launch each way (csrc/tokens.hip pcm_act_loss_*): replaces ~27 framework launches on tensors of
# a few hundred elements on the critical path between forward and backward.
class _ActLoss(Function):
    @staticmethod
    def forward(ctx, a_hat, actions, is_pad, mu, logvar, kl_weight):
        L = _lib.load()
        dev = a_hat.device
        B, D = mu.shape
        n, A = a_hat.numel(), a_hat.shape[-1]
        with torch.cuda.device(dev):
            ws = torch.empty(3 + n + 2 * B * D, dtype=torch.float32, device=dev)
            stats, ga, gmu, glv = ws[:3], ws[3:3 + n], ws[3 + n:3 + n + B * D], ws[3 + n + B * D:]
            rc = L.pcm_act_loss_forward_hip(n, A, B * D, B, int(a_hat.dtype == torch.bfloat16), a_hat.data_ptr(), actions.data_ptr(),
                                            is_pad.data_ptr(), int(mu.dtype == torch.bfloat16), mu.data_ptr(), logvar.data_ptr(),
                                            float(kl_weight), stats.data_ptr(), ga.data_ptr(), gmu.data_ptr(), glv.data_ptr(), _raw_stream())
        _lib.check(rc, "pcm_act_loss_forward_hip")
        ctx.set_materialize_grads(False)  # action_loss / kl_loss are logged, not differentiated: no zero gradients for them
        ctx.save_for_backward(ws)
        ctx.meta = (n, B * D, float(kl_weight), a_hat.shape, a_hat.dtype, mu.shape, mu.dtype)
        return stats[0], stats[1], stats[2]

    @staticmethod
    def backward(ctx, g_loss, g_action, g_kl):
        L = _lib.load()
        (ws,) = ctx.saved_tensors
        n, bd, kl_weight, a_shape, a_dtype, mu_shape, mu_dtype = ctx.meta
        dev = ws.device
        with torch.cuda.device(dev):
            da = torch.empty(a_shape, dtype=a_dtype, device=dev)
            dml = torch.empty((2,) + tuple(mu_shape), dtype=mu_dtype, device=dev)
            gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_loss, g_action, g_kl)]
            rc = L.pcm_act_loss_backward_hip(n, bd, *(0 if g is None else g.data_ptr() for g in gs), kl_weight, ws[3:].data_ptr(),
                                             ws[3 + n:].data_ptr(), ws[3 + n + bd:].data_ptr(), int(a_dtype == torch.bfloat16), da.data_ptr(),
                                             int(mu_dtype == torch.bfloat16), dml[0].data_ptr(), dml[1].data_ptr(), _raw_stream())
        _lib.check(rc, "pcm_act_loss_backward_hip")
        return da, None, None, dml[0], dml[1], None


def act_loss_supported(a_hat, actions, is_pad, mu, logvar, action_loss, klloss):
    from .losses import KLDivergence

    return (current() is not None and a_hat.is_cuda and type(action_loss) is torch.nn.MSELoss and action_loss.reduction == "none"
            and type(klloss) is KLDivergence and mu is not None and mu.dim() == 2 and a_hat.dim() == 3
            and a_hat.dtype in (torch.float32, torch.bfloat16) and mu.dtype in (torch.float32, torch.bfloat16) and mu.dtype == logvar.dtype
            and actions.dtype == torch.float32 and actions.shape == a_hat.shape and is_pad.dtype == torch.bool
            and is_pad.shape == a_hat.shape[:2] and torch.is_grad_enabled())


def act_loss(a_hat, actions, is_pad, mu, logvar, kl_weight):
    """-> (loss, action_loss, kl_loss), the three scalars of ACT.forward_loss."""
    return _ActLoss.apply(a_hat.contiguous(), actions.contiguous(), is_pad.contiguous().view(torch.uint8), mu.contiguous(),
                          logvar.contiguous(), kl_weight)


# ----------------------------------------------------------------------------------------------------------------------
# CVAE latent head in one launch each way (csrc/tokens.hip pcm_cvae_latent_*): split, reparametrisation, contiguous mu /
# logvar for the KL term -- and no framework RNG inside the captured step (the noise comes from the dropout masks' counter hash).
class _CVAELatent(Function):
    @staticmethod
    def forward(ctx, info, eps, seed, site):
        L = _lib.load()
        B, D2 = info.shape
        D = D2 // 2
        dev = info.device
        info = info.contiguous()
        bf = info.dtype == torch.bfloat16
        with torch.cuda.device(dev):
            z = torch.empty(B, D, dtype=torch.float32, device=dev)
            mu = torch.empty(B, D, dtype=info.dtype, device=dev)
            lv = torch.empty(B, D, dtype=info.dtype, device=dev)
            ws = torch.empty(2, B, D, dtype=torch.float32, device=dev)  # eps | std
            e = eps.float().contiguous() if eps is not None else None
            rc = L.pcm_cvae_latent_forward_hip(B, D, int(bf), info.data_ptr(), e.data_ptr() if e is not None else 0,
                                               seed.data_ptr() if seed is not None else 0, int(site), z.data_ptr(), mu.data_ptr(),
                                               lv.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), _raw_stream())
        _lib.check(rc, "pcm_cvae_latent_forward_hip")
        ctx.save_for_backward(ws)
        ctx.meta = (B, D, info.dtype)
        ctx.set_materialize_grads(False)
        return z, mu, lv

    @staticmethod
    def backward(ctx, dz, dmu, dlv):
        (ws,) = ctx.saved_tensors
        B, D, dt = ctx.meta
        if dz is None and dmu is None and dlv is None:
            return None, None, None, None
        L = _lib.load()
        dev = ws.device
        with torch.cuda.device(dev):
            dz = dz.float().contiguous() if dz is not None else None
            dmu = dmu.to(dt).contiguous() if dmu is not None else None
            dlv = dlv.to(dt).contiguous() if dlv is not None else None
            dinfo = torch.empty(B, 2 * D, dtype=dt, device=dev)
            rc = L.pcm_cvae_latent_backward_hip(B, D, int(dt == torch.bfloat16), dz.data_ptr() if dz is not None else 0,
                                                dmu.data_ptr() if dmu is not None else 0, dlv.data_ptr() if dlv is not None else 0,
                                                ws[0].data_ptr(), ws[1].data_ptr(), dinfo.data_ptr(), _raw_stream())
        _lib.check(rc, "pcm_cvae_latent_backward_hip")
        return dinfo, None, None, None


def cvae_latent_supported(info, latent_dim, eps):
    return (_ACTIVE is not None and info.is_cuda and info.dim() == 2 and info.shape[1] == 2 * latent_dim
            and info.dtype in (torch.float32, torch.bfloat16) and (eps is None or (eps.is_cuda and tuple(eps.shape) == (info.shape[0], latent_dim))))


def cvae_latent(info, eps=None):
    """(latent_sample fp32, mu, logvar) from latent_info = [mu | logvar]; eps None: noise from the context's counter hash."""
    ctx = _ACTIVE
    return _CVAELatent.apply(info, eps, ctx.seed if eps is None else None, ctx.next_site())

