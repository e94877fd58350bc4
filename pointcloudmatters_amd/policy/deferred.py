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
second gradient).  Outside a window ``push`` returns False and the producer launches its own reduction."""
import ctypes

import torch

from .. import _lib
from .._lib import raw_stream as _raw_stream

_Q = None  # None: no window.  Else the pending reductions of this backward stage.
STATS = {"pushed": 0, "launches": 0}  # tests / tools read these


def active():
    return _Q is not None


def begin():
    global _Q
    if _Q is None:
        _Q = []
        return True
    return False


def end():
    global _Q
    try:
        flush()
    finally:
        _Q = None


def targets(*params):
    """Forward-time half of the check: (ok, leaves).  ok = every tensor's gradient goes to the optimizer untouched (a leaf
    parameter or the trainer's bf16 mirror of one, rows_linear.goes_to_optimizer); leaves = the leaf parameters among them."""
    from .rows_linear import goes_to_optimizer

    ps = [p for p in params if p is not None]
    return all(goes_to_optimizer(p) for p in ps), [p for p in ps if p.grad_fn is None and p.requires_grad]


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


def flush():
    """Launch the pending reductions on the current stream (capturable: the table travels as a kernel argument)."""
    global _Q
    if not _Q:
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
    return n
