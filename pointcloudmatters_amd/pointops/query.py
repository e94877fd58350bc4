"""knn_query / ball_query / random_ball_query -- mirrors
/root/reference/libs/pointops/functions/query.py:6-112 (same argument order, same outputs:
idx (m, nsample) int32 with -1 placeholders, and sqrt(dist2))."""
import torch

from . import _common as C


def _prep(xyz, offset, new_xyz, new_offset):
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    C.require_hip(xyz, new_xyz, offset, new_offset)
    C.f32c(xyz, "xyz")
    C.f32c(new_xyz, "new_xyz")
    return new_xyz, new_offset


def knn_query_raw(nsample, xyz, offset, new_xyz=None, new_offset=None):
    """(idx, dist2): the native outputs, before the wrapper's sqrt (query.py:23)."""
    new_xyz, new_offset = _prep(xyz, offset, new_xyz, new_offset)
    L = C.lib()
    m = new_xyz.shape[0]
    with torch.cuda.device(xyz.device):
        idx = torch.empty(m, nsample, dtype=torch.int32, device=xyz.device)
        dist2 = torch.empty(m, nsample, dtype=torch.float32, device=xyz.device)
        o32, no32 = C.i32c(offset), C.i32c(new_offset)
        # n_max is only a hint (reserved by the launcher): use the host copy when it rides on the tensor, never read the device
        # for it (that would be a sync per call, and illegal under stream capture)
        host = getattr(offset, "_pcm_host", None)
        n_max = max(C.counts_from_offsets(host), default=0) if host else 0
        rc = L.pcm_knn_query_n_hip(
            int(offset.shape[0]), int(n_max), m, nsample, C.ptr(xyz), C.ptr(new_xyz), C.ptr(o32), C.ptr(no32), C.ptr(idx),
            C.ptr(dist2), C.stream(),
        )
    C._lib.check(rc, "pcm_knn_query_hip")
    return idx, dist2


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    idx, dist2 = knn_query_raw(nsample, xyz, offset, new_xyz, new_offset)
    return idx, torch.sqrt(dist2)


def ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    new_xyz, new_offset = _prep(xyz, offset, new_xyz, new_offset)
    assert min_radius < max_radius
    L = C.lib()
    m = new_xyz.shape[0]
    with torch.cuda.device(xyz.device):
        idx = torch.empty(m, nsample, dtype=torch.int32, device=xyz.device)
        dist2 = torch.empty(m, nsample, dtype=torch.float32, device=xyz.device)
        o32, no32 = C.i32c(offset), C.i32c(new_offset)
        # two kernels and a workspace where pcm_ball_query_ws_bytes(m) > 0 (from 8192 queries on: candidate collection at full
        # occupancy, then the heap replay, csrc/ball.hip); the single-kernel path below that
        nbytes = int(L.pcm_ball_query_ws_bytes(m))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device) if nbytes else None
        rc = L.pcm_ball_query_ws_hip(
            int(no32.shape[0]), m, nsample, float(min_radius), float(max_radius), C.ptr(xyz), C.ptr(new_xyz), C.ptr(o32), C.ptr(no32),
            C.ptr(idx), C.ptr(dist2), ws.data_ptr() if ws is not None else 0, nbytes, C.stream(),
        )
    C._lib.check(rc, "pcm_ball_query_ws_hip")
    return idx, dist2


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    idx, dist2 = ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset)
    return idx, torch.sqrt(dist2)


def make_random_order(offset):
    """query.py:46-53: per-cloud torch.randperm (int32, on the offsets' device) + cloud start."""
    host = C.host_offsets(offset)
    order, start = [], 0
    for e in host:
        order.append(torch.randperm(e - start, dtype=torch.int32, device=offset.device) + start)
        start = e
    return torch.cat(order, dim=0)


def random_ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None):
    new_xyz, new_offset = _prep(xyz, offset, new_xyz, new_offset)
    assert min_radius < max_radius
    L = C.lib()
    m = new_xyz.shape[0]
    if order is None:
        order = make_random_order(offset)
    C.require_hip(order)
    order = C.i32c(order)
    with torch.cuda.device(xyz.device):
        idx = torch.empty(m, nsample, dtype=torch.int32, device=xyz.device)
        dist2 = torch.empty(m, nsample, dtype=torch.float32, device=xyz.device)
        o32, no32 = C.i32c(offset), C.i32c(new_offset)
        rc = L.pcm_random_ball_query_b_hip(
            int(no32.shape[0]), m, nsample, float(min_radius), float(max_radius), C.ptr(order), C.ptr(xyz), C.ptr(new_xyz), C.ptr(o32),
            C.ptr(no32), C.ptr(idx), C.ptr(dist2), C.stream(),
        )
    C._lib.check(rc, "pcm_random_ball_query_b_hip")
    return idx, dist2


def random_ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None):
    """``order`` is an extension (the reference draws it internally): pass it for reproducible parity."""
    idx, dist2 = random_ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset, order)
    return idx, torch.sqrt(dist2)
