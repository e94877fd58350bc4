"""Closing reductions of a backward stage, batched (csrc/tokens.hip: pcm_reduce_batch_hip).

The fused backward kernels (drln, ffn_ln, the in-projection's bias sums, split-K weight gradients) end in a second-level
reduction ``out[e] = sum_s partial[s][e]``: 70+ launches of 5-7 us each in one ACT step, a twelfth of its launches.  Nothing
reads those results before the optimizer's gradient hand-off, so a training loop that hands gradients over explicitly
(bc/trainer.py, "collect" mode) opens a window with ``begin()``; inside it the producers ``push`` their reduction instead of
launching it and return the (still unwritten) result tensor to autograd, and ``flush()`` -- called after every backward
stage, before anything reads a gradient -- closes all of them with one launch per 24.  Same arithmetic and order as the
per-reduction kernels: results are bit-identical with and without the window (tests/test_policy_gpu.py).

Rules the producers keep: the result tensors must reach an AccumulateGrad node / the optimizer's stash untouched (no cast, no
add); anything that has to READ a pending result first calls ``flush()`` (flat_optim.stash_grad does when a weight receives a
second gradient).  Outside a window ``push`` returns False and the producer launches its own reduction.

Weight gradients ride the same window (``push_wgrad``): dW = dY^T X of a short activation (the decoder's 800 rows) is one
latency-bound product per layer (11 us for 0.4 GFLOP on 64 workgroups), a leaf of the backward graph like the reductions.
Products of one shape are collected over the stage and run as ONE batched product at the flush (7 layers: 80 us -> 21 us
for the product itself, plus two stacking copies).  Unlike the reductions this changes the summation order inside the GEMM
(another hipBLASLt kernel): results agree to rounding, not bit for bit (BATCH_WGRADS = False restores the single products)."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import raw_stream as _raw_stream

_Q = None  # None: no window.  Else the pending reductions of this backward stage.
_W = {}    # pending weight gradients of this stage by shape key
_SEEN = {}    # shape key -> products computed singly since the last flush
_EXPECT = {}  # shape key -> products seen at the last flush (sizes the shared result buffer of the next stage with that key)
_DEBUG = os.environ.get("PCM_DEFER_DEBUG", "0") != "0"
BATCH_WGRADS = os.environ.get("PCM_BATCH_WGRADS", "1") != "0"
WGRAD_MIN_GROUP = int(os.environ.get("PCM_WGRAD_MIN_GROUP", 4))  # fewer products of a shape than this are computed singly (C2: 5.63 ms with 5, 5.55 with 4)
STATS = {"pushed": 0, "launches": 0, "wgrads": 0, "wgrad_batches": 0, "stacked": 0, "scattered": 0}  # tests / tools read these


def active():
    return _Q is not None


def begin():
    global _Q
    if _Q is None:
        _Q = []
        return True
    return False


def end(failed=False):
    """Close the window.  `failed=True` (the caller's body raised, or its generator was closed early): the queues are half-built,
    drop them without a closing flush so that the original exception is not masked.  The caller says so explicitly -- the ambient
    `sys.exc_info()` is also set when a healthy step runs inside someone else's `except` block (an out-of-memory retry loop)."""
    global _Q, _W

    try:
        if not failed:
            flush()
    finally:
        _Q = None
        _W = {}
        _C.clear()
        _P.clear()
        _SEEN.clear()
        _ARENAS.clear()
        _TAKE_EXPECT.clear()
        _TAKE_EXPECT.update(_TAKEN)
        _TAKEN.clear()


def handout(t):
    """The tensor a producer returns to autograd for a PENDING result `t` that the queues still reference: a fresh alias.
    AccumulateGrad takes a gradient without copying only when nothing else references that tensor object -- and a view
    kept in a queue references its base -- otherwise it CLONES it on the spot, i.e. reads the result before it is written.
    So the queues keep `t` (and views of it), autograd gets this separate alias."""
    return t.view(t.shape)


def targets(*params):
    """Forward-time half of the check: (ok, leaves).  ok = every tensor's gradient goes to the optimizer untouched (a leaf
    parameter or the trainer's bf16 mirror of one, rows_linear.goes_to_optimizer); leaves = the leaf parameters among them."""
    from .rows_linear import goes_to_optimizer

    ps = [p for p in params if p is not None]
    return all(goes_to_optimizer(p) for p in ps), [p for p in ps if p.is_leaf and p.requires_grad]


def clear(ok, leaves):
    """Backward-time half: may this node leave its reductions pending?  A leaf that already holds a gradient would have
    the new one ADDED to it by AccumulateGrad (a read): flush what is pending and reduce immediately instead."""
    if _Q is None or not ok:
        return False
    for p in leaves:
        if p.grad is not None:
            flush()
            return False
    return True


def push(partial, nslots, width, out_f32=None, out_bf16=None, bf16_from=0):
    """Queue out = sum over the nslots rows of `partial` (nslots * width fp32).  The queue keeps ALIASES of the tensors (their
    storage must outlive the flush; the tensor objects handed to autograd must stay uniquely referenced so AccumulateGrad
    can take them without a copy)."""
    if _Q is None or not partial.is_cuda or width <= 0:
        return False
    keep = (partial, out_f32.view(-1) if out_f32 is not None else None, out_bf16.view(-1) if out_bf16 is not None else None)
    _Q.append((partial.data_ptr(), int(nslots), int(width), out_f32.data_ptr() if out_f32 is not None else 0,
               out_bf16.data_ptr() if out_bf16 is not None else 0, int(bf16_from), partial.device, keep))
    STATS["pushed"] += 1
    return True


# ---- operand arenas ------------------------------------------------------------------------------------------------
# A batched product wants its operands as ONE strided tensor.  The fused nodes allocate the activations that later become
# weight-gradient operands (attention outputs, dy of the norms, the bf16 inputs of the projections) through ``take``:
# inside a window, tensors of one (shape, dtype) are consecutive slots of one buffer, sized from the count of the previous
# window.  Identical layers allocate in an identical order, so the l-th layer's operand sits at a fixed slot stride and the
# flush can describe all of them with as_strided -- no stacking copy.  Anything else (first step, other allocation order,
# tensors from framework ops) falls back to torch.stack.
_ARENAS = {}       # (shape, dtype, device) -> [buffer, next slot]
_TAKEN = {}        # takes per key in this window
_TAKE_EXPECT = {}  # ... in the previous window
ARENA_MAX_BYTES = 16 << 20  # per slot (and per operand of a batched product): larger activations keep the split-K route


_BACKWARD = False  # set by the training loop around its autograd calls (``backward_phase``)


class backward_phase:
    """with deferred.backward_phase(): torch.autograd.backward(...) -- tells ``take`` which way to fill its arenas."""

    def __enter__(self):
        global _BACKWARD
        self.prev, _BACKWARD = _BACKWARD, True

    def __exit__(self, *exc):
        global _BACKWARD
        _BACKWARD = self.prev


def take(shape, dtype, device, tag=None):
    """torch.empty(shape) -- as a slot of this window's arena for that (call site `tag`, shape) when weight gradients are
    being batched: one site's tensors are consecutive slots whatever else is allocated in between.
    Forward allocations fill their arena upwards, backward allocations fill theirs DOWNWARDS: backward visits the layers in
    reverse, so in the order the weight gradients are pushed both kinds of operand then sit at descending addresses, and
    the flush (which batches in reverse push order) sees two ascending, uniformly strided sets."""
    shape = tuple(int(v) for v in shape)
    if _Q is None or not BATCH_WGRADS:
        return torch.empty(shape, dtype=dtype, device=device)
    key = (tag, shape, dtype, torch.device(device), _BACKWARD)
    _TAKEN[key] = _TAKEN.get(key, 0) + 1
    ar = _ARENAS.get(key)
    if ar is None:
        n = _TAKE_EXPECT.get(key, 0)
        numel = 1
        for v in shape:
            numel *= v
        if n < WGRAD_MIN_GROUP or numel * torch.empty((), dtype=dtype).element_size() > ARENA_MAX_BYTES or numel == 0:
            return torch.empty(shape, dtype=dtype, device=device)
        # a quarter more slots than needed, free at the top: the padded batch (7 -> 8) of a strided operand set reads one
        # stride past its last member and stays inside the buffer
        ar = _ARENAS[key] = [torch.empty((n + (n + 3) // 4,) + shape, dtype=dtype, device=device), 0, n]
    if ar[1] >= ar[2]:
        return torch.empty(shape, dtype=dtype, device=device)
    t = ar[0][ar[2] - 1 - ar[1] if _BACKWARD else ar[1]]
    ar[1] += 1
    return t


def _extent_bytes(t):
    """Bytes from t's first to one past its last element, for the layouts a batched product can address: contiguous, or a
    2-D row-strided matrix (a column block of a wider buffer: dq | dk of a joint dq | dk | dv gradient).  None otherwise."""
    es = t.element_size()
    if t.is_contiguous():
        return t.numel() * es
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.shape[0] > 0:
        return ((t.shape[0] - 1) * t.stride(0) + t.shape[1]) * es
    return None


def _as_batch(ts, nb):
    """The tensors of `ts` (same shape, contiguous) as one (n', ...) strided view when they sit at a uniform stride in one
    storage: n' = nb if that many slots fit in the storage, else len(ts).  None when they do not line up."""
    t0 = ts[0]
    es, p0 = t0.element_size(), t0.data_ptr()
    if len(ts) < 2:
        return None
    d = ts[1].data_ptr() - p0
    ext = _extent_bytes(t0)
    if ext is None or d <= 0 or d % es or d < ext or any(t.stride() != t0.stride() for t in ts):
        return None
    st = t0.untyped_storage()
    base = st.data_ptr()
    for i, t in enumerate(ts):
        if t.data_ptr() - p0 != i * d or t.untyped_storage().data_ptr() != base:
            return None
    n = nb if p0 + (nb - 1) * d + ext <= base + st.nbytes() else len(ts)
    return torch.as_strided(t0, (n,) + tuple(t0.shape), (d // es,) + tuple(t0.stride()))


def _pad_is_free(ov, n):
    """May a batched product write the padding batches ov[n:] of a strided OUTPUT?  Only when they are never-handed-out slots
    of one of this window's arenas (`take` keeps a quarter more slots than it hands out, at the top).  Anything else -- a
    caller's slice of a flat / packed gradient, or arena slots that an earlier flush of this stage already filled and gave
    to autograd -- must not be touched: the padded batch would rewrite somebody's finished gradient."""
    es = ov.element_size()
    step = ov.stride(0) * es
    ext = _extent_bytes(ov[0])
    if ext is None or step <= 0:
        return False
    lo, hi = ov.data_ptr() + n * step, ov.data_ptr() + (ov.shape[0] - 1) * step + ext
    base = ov.untyped_storage().data_ptr()
    for buf, _used, cap in _ARENAS.values():
        if buf.untyped_storage().data_ptr() == base:
            slot = buf.stride(0) * buf.element_size()
            return lo >= buf.data_ptr() + cap * slot and hi <= buf.data_ptr() + buf.shape[0] * slot
    return False


def _padded(n):
    # hipBLASLt's batched kernels are markedly faster at 4 / 8 / 16 batches than at 5, 7 or 14 (800 x 512 x 512 bf16: 14
    # batches 24.7 us, 16 batches 16.9 us; tools/mb/mb_wgrad_batch.py, mb_wgrad_enc.py)
    for p, slack in ((4, 1), (8, 1), (16, 2)):
        if n <= p and p - n <= slack:
            return p
    return n


def push_wgrad(go, x, out_dtype, out=None, tag=None):
    """Queue dW = go^T @ x (go (rows, m), x (rows, k), contiguous, same dtype) and return the tensor that WILL hold it (`out`,
    a contiguous (m, k) tensor or slice, when given), or None when the caller should compute it now (no window, batching
    off, or a shape that came fewer than WGRAD_MIN_GROUP times in the last stage that had it)."""
    if _Q is None or not BATCH_WGRADS or not go.is_cuda or go.dtype != x.dtype or go.dtype not in (torch.bfloat16, torch.float32) \
            or go.dim() != 2 or _extent_bytes(go) is None or not x.is_contiguous() or (out is not None and not out.is_contiguous()):
        return None
    if go.dtype == torch.float32 and out_dtype != torch.float32:
        return None
    if max(go.numel(), x.numel()) * go.element_size() > ARENA_MAX_BYTES:
        return None
    key = (tuple(go.shape), tuple(x.shape), go.dtype, out_dtype, go.device, tag, tuple(go.stride()))
    want = _EXPECT.get(key, 0)
    grp = _W.get(key)
    if grp is None:
        if want < WGRAD_MIN_GROUP:
            _SEEN[key] = _SEEN.get(key, 0) + 1  # computed singly, counted: the next stage with this many of them batches
            return None
        grp = _W[key] = {"buf": None, "gos": [], "xs": [], "extra": []}
    i = len(grp["gos"])
    if out is None and i < want:
        if grp["buf"] is None:
            grp["buf"] = torch.empty(_padded(want), go.shape[1], x.shape[1], dtype=out_dtype, device=go.device)
        dw = grp["buf"][want - 1 - i]  # the flush batches in REVERSE push order (see take)
    else:  # a caller's slice, or more products than last time: filled by the scatter copy
        dw = out if out is not None else torch.empty(go.shape[1], x.shape[1], dtype=out_dtype, device=go.device)
        grp["extra"].append((i, dw.view(go.shape[1], x.shape[1])))
        if out is None:
            dw = handout(dw)
    grp["gos"].append(go)
    grp["xs"].append(x)
    STATS["wgrads"] += 1
    return dw


def _flush_wgrads():
    global _W, _SEEN
    if _SEEN:
        _EXPECT.update(_SEEN)
        _SEEN = {}
    if not _W:
        return
    groups, _W = _W, {}
    for key, grp in groups.items():
        gos, xs, buf = grp["gos"][::-1], grp["xs"][::-1], grp["buf"]  # batch j = push n-1-j
        n = len(gos)
        want = _EXPECT.get(key, 0)
        _EXPECT[key] = n
        out_dtype = key[3]
        nb = _padded(n)
        a, b = _as_batch(gos, nb), _as_batch(xs, nb)
        if a is not None and b is not None and a.shape[0] != b.shape[0]:
            nb = n
            a, b = a[:n], b[:n]
        elif a is not None or b is not None:
            nb = (a if a is not None else b).shape[0]
        STATS["stacked"] += (a is None) + (b is None)
        if _DEBUG and (a is None or b is None):
            print("deferred: stacked", "go" if a is None else "", "x" if b is None else "", key[:4],
                  [t.data_ptr() - gos[0].data_ptr() for t in gos], [t.data_ptr() - xs[0].data_ptr() for t in xs], flush=True)
        if a is None:
            a = torch.stack(gos + gos[: nb - n])
        if b is None:
            b = torch.stack(xs + xs[: nb - n])
        kw = {"out_dtype": out_dtype} if (a.dtype == torch.bfloat16 and out_dtype == torch.float32) else {}
        ov = None
        if len(grp["extra"]) == n and not kw:  # every product has its own destination (slices of packed gradients taken
            ov = _as_batch([d for _, d in grp["extra"]][::-1], nb)  # from an arena): one strided output if they line up
            if ov is not None and ov.shape[0] > n and not _pad_is_free(ov, n):
                ov = ov[:n]  # the slots behind the last destination are not ours to write: unpadded product
            if ov is not None and ov.shape[0] not in (n, nb):
                ov = None
        if ov is not None:
            k = ov.shape[0]
            torch.bmm(a[:k].transpose(1, 2), b[:k], out=ov)
        elif not grp["extra"] and n == want and buf is not None and buf.shape[0] == nb and not kw:
            torch.bmm(a.transpose(1, 2), b, out=buf)  # push i was handed buf[want-1-i] = batch n-1-i
        else:
            STATS["scattered"] += 1
            r = torch.bmm(a.transpose(1, 2), b, **kw)
            taken = {i for i, _ in grp["extra"]}
            own = [i for i in range(min(n, want)) if i not in taken]  # push indices that were handed a slot of buf
            dsts = [buf[want - 1 - i].view(-1) for i in own] + [d.view(-1) for _, d in grp["extra"]]
            srcs = [r[n - 1 - i].reshape(-1) for i in own] + [r[n - 1 - i].reshape(-1) for i, _ in grp["extra"]]
            torch._foreach_copy_(dsts, srcs)
        STATS["wgrad_batches"] += 1


_C = []  # pending first stages of column sums (their closing stages are already in _Q)


def push_colsum(rows, C, tensors, lds, in_dtype, partial, keep):
    """Queue the FIRST stage of a column sum (pcm_colsum_hip's arguments: up to three (pointer, row stride) pairs of one
    dtype); the caller has pushed its closing stage.  `keep`: tensors that must outlive the flush."""
    if _Q is None:
        return False
    g = [int(p) for p in tensors] + [0] * (3 - len(tensors))
    ld = [int(v) for v in lds] + [0] * (3 - len(lds))
    _C.append((int(rows), int(C), len(tensors), int(in_dtype == torch.bfloat16), g, ld, partial.data_ptr(), partial.device, (partial, keep)))
    return True


def _flush_colsums():
    global _C
    if not _C:
        return
    q, _C = _C, []
    n = len(q)
    Lg, I, P = ctypes.c_long * n, ctypes.c_int * n, ctypes.c_void_p * n
    g = (ctypes.c_void_p * (3 * n))(*[p or None for e in q for p in e[4]])
    ld = (ctypes.c_long * (3 * n))(*[v for e in q for v in e[5]])
    with torch.cuda.device(q[0][7]):
        rc = _lib.load().pcm_colsum_batch_hip(n, Lg(*[e[0] for e in q]), I(*[e[1] for e in q]), I(*[e[2] for e in q]),
                                              I(*[e[3] for e in q]), g, ld, P(*[e[6] for e in q]), _raw_stream())
    _lib.check(rc, "pcm_colsum_batch_hip")
    STATS["launches"] += (n + 15) // 16


_P = []  # pending copies (dst, src): run LAST in a flush, after everything that produces a src


def push_copies(pairs):
    """Queue dst.copy_(src) for every pair (same shape and dtype per pair); all pending copies of a stage are one multi-tensor
    launch per dtype.  The sources may themselves be pending results of this window."""
    if _Q is None:
        return False
    _P.extend(pairs)
    return True


def _flush_copies():
    global _P
    if not _P:
        return
    q, _P = _P, []
    _lib.copy_pairs(q)  # same-dtype contiguous pairs: one table-driven launch per 96 (csrc/optim.hip); the rest: multi-tensor copies
    STATS["launches"] += 1


@torch.no_grad()
def flush():
    """Launch the pending reductions on the current stream (capturable: the table travels as a kernel argument)."""
    global _Q
    _flush_wgrads()
    _flush_colsums()
    if not _Q:
        _flush_copies()
        return 0
    q, _Q = _Q, []
    n = len(q)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    dev = q[0][6]
    with torch.cuda.device(dev):
        rc = _lib.load().pcm_reduce_batch_hip(n, P(*[e[0] for e in q]), I(*[e[1] for e in q]), I(*[e[2] for e in q]),
                                              P(*[e[3] or None for e in q]), P(*[e[4] or None for e in q]), I(*[e[5] for e in q]),
                                              _raw_stream())
    _lib.check(rc, "pcm_reduce_batch_hip")
    STATS["launches"] += (n + 23) // 24
    _flush_copies()
    return n
