"""farthest_point_sampling -- mirrors /root/reference/libs/pointops/functions/sampling.py:6-26."""
import torch

from . import _common as C


def farthest_point_sampling(xyz, offset, new_offset):
    """input: xyz (n, 3) float32, offset (b), new_offset (b); output: idx (m) int32, global indices.

    Same contract as FarthestPointSampling.forward (sampling.py:8-24): first pick of every cloud is
    its first point; n_max = max cloud size selects the reference block size that defines tie order.
    """
    assert xyz.is_contiguous()
    C.require_hip(xyz, offset, new_offset)
    C.f32c(xyz, "xyz")
    L = C.lib()
    off_h = C.host_offsets(offset)
    noff_h = C.host_offsets(new_offset)
    b = len(off_h)
    n_max = max(C.counts_from_offsets(off_h)) if b else 0
    m = noff_h[b - 1] if b else 0
    with torch.cuda.device(xyz.device):
        idx = torch.zeros(m, dtype=torch.int32, device=xyz.device)
        if b == 0 or m == 0:
            return idx
        # tmp (sampling.py:18) is only touched by the > 16384-points-per-cloud kernel
        tmp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device) if n_max > 16384 else None
        o32, no32 = C.i32c(offset), C.i32c(new_offset)
        rc = L.pcm_farthest_point_sampling_hip(
            b, n_max, C.ptr(xyz), C.ptr(o32), C.ptr(no32), C.ptr(tmp), C.ptr(idx), C.stream()
        )
    C._lib.check(rc, "pcm_farthest_point_sampling_hip")
    return idx
