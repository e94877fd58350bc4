"""Shared plumbing for the pointops wrappers: HIP-tensor checks, stream, host-side offsets."""
import torch

from .. import _lib
from .._lib import raw_stream as _raw_stream


def lib():
    return _lib.load()


def require_hip(*tensors):
    """Every product op runs on the GPU through libpcm_pointops.so; anything else is an error
    (the reference has no CPU path either: its wrappers allocate torch.cuda.*Tensor)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.PointopsLibraryError(
                "pointops: expected a HIP (torch 'cuda') tensor, got device=%s. There is no CPU fallback." % t.device
            )


def f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"pointops: {name} must be float32, got {t.dtype}")
    assert t.is_contiguous(), f"pointops: {name} must be contiguous"
    return t


def i32c(t):
    """offset.int() of the reference wrappers (sampling.py:20, query.py:21): no-op when already int32."""
    t = t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return _raw_stream()


def host_offsets(offset):
    """Cumulative offsets as a Python list.

    The reference reads offsets element by element from the device (b+1 host syncs in
    functions/sampling.py:14-17).  Here a tensor produced by our collate carries its host copy in
    the attribute ``_pcm_host``; otherwise ONE device->host copy is made and cached on the tensor.
    """
    h = getattr(offset, "_pcm_host", None)
    if h is None:
        h = [int(v) for v in offset.tolist()]
        try:
            offset._pcm_host = h
        except Exception:  # pragma: no cover - tensors always accept attributes
            pass
    return h


def with_host(offset, host):
    """Attach a known host copy to a device offset tensor (no sync later)."""
    offset._pcm_host = [int(v) for v in host]
    return offset


def counts_from_offsets(host):
    return [host[0]] + [host[i] - host[i - 1] for i in range(1, len(host))]


class ScatterPlan:
    """CSR inverse of an index list (csrc/segsum.hip): which entries of ``idx`` land on each of the ``n_dst``
    destination rows.  Built once per backward with five small launches; the scatter-adds of grouping /
    interpolation / subtraction / aggregation backward then become atomic-free segmented sums."""

    def __init__(self, idx, n_dst):
        import ctypes

        L = lib()
        self.rows, self.n_dst = idx.numel(), int(n_dst)
        self.idx = idx
        with torch.cuda.device(idx.device):
            self.ws = torch.empty(L.pcm_scatter_plan_ws_ints(self.rows, self.n_dst), dtype=torch.int32, device=idx.device)
            start, lst = ctypes.c_void_p(), ctypes.c_void_p()
            rc = L.pcm_scatter_plan_hip(self.rows, self.n_dst, ptr(idx), ptr(self.ws), ctypes.byref(start), ctypes.byref(lst), stream())
        _lib.check(rc, "pcm_scatter_plan_hip")
        self.start, self.list = start.value or 0, lst.value or 0


def segment_sum(dst, src, *, src_stride=None, src_off=0, plan=None, seglen=0, map=None, rowdiv=1, scale=None, scale_mode=0,
                w_c=1, sign=1.0):
    """dst (n_dst, c) = sign * segmented sum of rows of ``src`` (see include/pcm_pointops.h, pcm_segment_sum_hip)."""
    n_dst, c = dst.shape
    with torch.cuda.device(dst.device):
        rc = lib().pcm_segment_sum_hip(
            n_dst, c, plan.start if plan is not None else 0, int(seglen), plan.list if plan is not None else 0, ptr(map),
            int(rowdiv), ptr(scale), int(scale_mode), int(w_c), float(sign), ptr(src),
            int(src_stride if src_stride is not None else c), int(src_off), ptr(dst), stream())
    _lib.check(rc, "pcm_segment_sum_hip")
    return dst
