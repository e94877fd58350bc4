"""GPU parity: HIP pointops (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bit-exact for FPS indices, kNN / ball-query neighbour lists and their dist2; fp32 gather outputs
exact; atomic-scatter backward within 1e-5 relative (summation order is free in the reference too).
"""
import pytest
import torch

from tests.util import make_clouds, new_offsets

pytestmark = pytest.mark.gpu

FPS_CASES = [
    # (sizes, ms, mode)
    ([512], [128], "uniform"),
    ([1024] * 8, [512] * 8, "uniform"),
    ([2048] * 4, [1024] * 4, "uniform"),
    ([4096] * 8, [2048] * 8, "uniform"),
    ([3100, 4096, 5000, 3600], [2048] * 4, "uniform"),       # ragged REF batch: 4096 < n_max <= 6144 -> 256 threads x 24 points
    ([6144, 4097], [2048, 700], "lattice"),                   # ... at its capacity, with forced ties
    ([8192, 7000, 6145], [1024, 512, 300], "uniform"),       # 6144 < n_max <= 8192 -> 256 threads x 32 points
    ([8000, 5000], [400, 400], "dup"),
    ([1000, 37, 260, 1023], [64, 50, 64, 600], "uniform"),    # ragged incl. N < M and N < BS
    ([5, 3, 1], [8, 2, 4], "uniform"),                        # tiny, M > N
    ([100], [100], "uniform"),                                # BS = 64 < T
    ([300, 511], [40, 300], "uniform"),                       # BS = 256
    ([700, 600], [350, 300], "uniform"),                      # BS = 512 (q = 2)
    ([1024] * 8, [512] * 8, "lattice"),                       # forced distance ties
    ([4096] * 2, [2048] * 2, "lattice"),
    ([1500, 900], [1024, 1024], "dup"),                       # duplicates + M > N for one cloud
    ([9000, 8000], [512, 512], "uniform"),                    # T = 1024, PPT = 16, xyz from global
    ([16384], [256], "lattice"),
    ([20000, 17000], [200, 100], "uniform"),                  # any-size kernel (tmp in HBM)
    ([1024] * 128, [512] * 128, "uniform"),                   # DP-style batch of 128 clouds
]


@pytest.mark.parametrize("sizes,ms,mode", FPS_CASES)
def test_fps_bit_exact(hip_device, sizes, ms, mode):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(sizes, seed=len(sizes) * 7 + sizes[0], mode=mode)
    noff = new_offsets(ms)
    want = ref.farthest_point_sampling(xyz, off, noff)
    got = po.farthest_point_sampling(xyz.to(hip_device), off.to(hip_device), noff.to(hip_device))
    assert got.dtype == torch.int32 and got.shape == want.shape
    assert torch.equal(got.cpu(), want), f"first mismatch at {(got.cpu() != want).nonzero()[:4].flatten().tolist()}"


KNN_CASES = [
    ([1024] * 8, [512] * 8, 16, "uniform"),
    ([4096] * 4, [2048] * 4, 16, "uniform"),
    ([1000, 37, 260, 1023], [64, 50, 64, 600], 16, "uniform"),
    ([5, 3, 20], [5, 3, 20], 16, "uniform"),                  # clouds smaller than nsample -> -1 padding
    ([1024] * 4, [512] * 4, 16, "lattice"),                   # exact ties -> exact kernel path
    ([2048] * 2, [1024] * 2, 16, "dup"),
    ([2000], [700], 3, "uniform"),                            # interpolation's k
    ([2000], [700], 63, "uniform"),                           # largest fast-path nsample
    ([2000], [300], 64, "uniform"),                           # exact kernel only
    ([500], [100], 128, "lattice"),
    ([300], [300], 1, "dup"),
    ([700, 300, 20, 5, 1030], [64, 64, 64, 64, 130], 16, "uniform"),   # query blocks of one workgroup straddling up to 4 clouds
    ([5000, 1], [2500, 3], 16, "lattice"),                    # > 8 chunks of 512 points; a one-point cloud behind it
    ([4096] * 8, [2048] * 8, 16, "uniform"),                  # m = 16384: four queries per wave
    ([1024] * 16, [600] * 16, 8, "dup"),                      # m = 9600: two queries per wave, duplicates
    # two-pass kernel: clouds of several 4096-point register stretches (buffer compacted in between), ragged stretch tails
    ([9000, 4097, 12289], [300, 200, 400], 16, "uniform"),
    ([8192 + 5], [500], 32, "lattice"),                       # ties across stretch boundaries -> exact kernel
    ([4500], [200], 62, "uniform"),                           # nsample + 1 = 63 list entries, candidates close to the buffer size
]


@pytest.mark.parametrize("sizes,ms,nsample,mode", KNN_CASES)
def test_knn_bit_exact(hip_device, sizes, ms, nsample, mode):
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd.pointops.query import knn_query_raw
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(sizes, seed=11 + nsample, mode=mode)
    noff = new_offsets(ms)
    sel = ref.farthest_point_sampling(xyz, off, noff).long()
    new_xyz = xyz[sel].contiguous()
    wi, wd = ref.knn_query_raw(nsample, xyz, off, new_xyz, noff)
    gi, gd = knn_query_raw(nsample, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device), noff.to(hip_device))
    assert torch.equal(gi.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)
    # public wrapper: sqrt(dist2), pads -> 1e5
    pi, pd = po.knn_query(nsample, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device), noff.to(hip_device))
    assert torch.equal(pi.cpu(), wi)
    # torch.sqrt on the GPU is not correctly rounded (1 ulp off the CPU's): fp32 feature tolerance
    torch.testing.assert_close(pd.cpu(), torch.sqrt(wd), rtol=1e-6, atol=0)


def test_knn_candidate_buffer_overflow_goes_to_the_exact_kernel(hip_device):
    """More than 128 points below the two-pass kernel's threshold: 400 copies of each of 8 locations.  Every lane minimum is
    one of 8 values, so the threshold admits hundreds of equal distances, the candidate buffer overflows and the query must
    come back from the exact kernel -- bit-exact like everything else (the reference's heap decides among the copies)."""
    from pointcloudmatters_amd.pointops.query import knn_query_raw
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(3)
    base = torch.rand(8, 3, generator=g)
    xyz = base.repeat_interleave(400, dim=0)[torch.randperm(3200, generator=g)].contiguous()
    xyz = torch.cat([xyz, torch.rand(900, 3, generator=g)]).contiguous()  # a second, ordinary cloud behind it
    off = torch.tensor([3200, 4100], dtype=torch.int32)
    new_xyz = torch.cat([base, torch.rand(24, 3, generator=g), xyz[3200:3232]]).contiguous()
    noff = torch.tensor([32, 64], dtype=torch.int32)
    wi, wd = ref.knn_query_raw(16, xyz, off, new_xyz, noff)
    gi, gd = knn_query_raw(16, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device), noff.to(hip_device))
    assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)


def test_knn_self_query_defaults(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds([700, 300], seed=5)
    wi, wd = ref.knn_query(8, xyz, off)
    gi, gd = po.knn_query(8, xyz.to(hip_device), off.to(hip_device))
    assert torch.equal(gi.cpu(), wi)
    torch.testing.assert_close(gd.cpu(), wd, rtol=1e-6, atol=0)


BALL_CASES = [
    ([1024] * 4, [256] * 4, 16, 0.1, 0.0, "uniform"),
    ([1024] * 2, [256] * 2, 32, 0.2, 0.02, "uniform"),       # cnt > nsample -> strided subsample + dist2:=idx quirk
    ([500, 40], [100, 40], 8, 0.05, 0.0, "lattice"),
    ([3000], [64], 16, 1.0, 0.0, "uniform"),                  # > 2048 candidates -> defined-empty row
    ([10, 3], [10, 3], 16, 0.5, 0.0, "uniform"),
]


@pytest.mark.parametrize("sizes,ms,nsample,rmax,rmin,mode", BALL_CASES)
def test_ball_query_bit_exact(hip_device, sizes, ms, nsample, rmax, rmin, mode):
    from pointcloudmatters_amd.pointops.query import ball_query_raw
    from oracle import lib as olib
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(sizes, seed=3, mode=mode)
    noff = new_offsets(ms)
    sel = ref.farthest_point_sampling(xyz, off, noff).long()
    new_xyz = xyz[sel].contiguous()
    L = olib.load()
    wi = torch.zeros(new_xyz.shape[0], nsample, dtype=torch.int32)
    wd = torch.zeros(new_xyz.shape[0], nsample, dtype=torch.float32)
    rc = L.pcm_ball_query_cpu(new_xyz.shape[0], nsample, rmin, rmax, xyz.data_ptr(), new_xyz.data_ptr(),
                              off.data_ptr(), noff.data_ptr(), wi.data_ptr(), wd.data_ptr())
    assert rc in (0, 2)  # 2 = some query overflowed 2048 candidates (UB in the reference)
    gi, gd = ball_query_raw(nsample, rmax, rmin, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device), noff.to(hip_device))
    assert torch.equal(gi.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)


BALL_MANY = [
    ([1024] * 64, [512] * 64, 16, 0.1, 0.0, "uniform"),                      # 32 768 queries: lane-per-query kernel, ~16 candidates each
    ([700, 1300, 2048, 333, 4096] * 5, [650, 1200, 2000, 300, 2500] * 5, 16, 0.12, 0.01, "lattice"),  # ragged, exact ties, 64-query blocks straddle clouds
    ([2048] * 20, [1700] * 20, 32, 0.25, 0.0, "uniform"),                    # > 96 candidates for most queries: flagged and redone wave-per-query
    ([70000, 500], [32500, 400], 8, 0.03, 0.0, "uniform"),                   # a cloud too large for 16-bit local indices
]


@pytest.mark.parametrize("sizes,ms,nsample,rmax,rmin,mode", BALL_MANY)
def test_ball_query_bit_exact_with_many_queries(hip_device, sizes, ms, nsample, rmax, rmin, mode):
    """>= 32 768 queries take the lane-per-query kernel (64 queries per wave, candidates in LDS columns, the heap_sort replay run
    for 64 queries at once); its flagged leftovers go through the wave-per-query kernel.  Same bits as the oracle either way."""
    from pointcloudmatters_amd.pointops.query import ball_query_raw
    from oracle import lib as olib

    xyz, off = make_clouds(sizes, seed=9, mode=mode)
    noff = new_offsets(ms)
    assert sum(ms) >= 32768
    starts = [0] + off.tolist()[:-1]
    sel = torch.cat([st + (torch.arange(mq) * 7919) % n for st, n, mq in zip(starts, sizes, ms)])  # queries: a strided subset of each cloud
    new_xyz = xyz[sel].contiguous()
    L = olib.load()
    wi = torch.zeros(new_xyz.shape[0], nsample, dtype=torch.int32)
    wd = torch.zeros(new_xyz.shape[0], nsample, dtype=torch.float32)
    rc = L.pcm_ball_query_cpu(new_xyz.shape[0], nsample, rmin, rmax, xyz.data_ptr(), new_xyz.data_ptr(),
                              off.data_ptr(), noff.data_ptr(), wi.data_ptr(), wd.data_ptr())
    assert rc in (0, 2)
    gi, gd = ball_query_raw(nsample, rmax, rmin, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device), noff.to(hip_device))
    assert torch.equal(gi.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)


BALL_SPLIT = [
    # the split path (collect + replay kernels, csrc/ball.hip) at every segment count S: m * S >= 262 144 picks S
    ([1100] * 8, [1024] * 8, 16, 0.1, 0.0, "uniform"),                        # 8 192 queries -> S = 8, segments of ~137 points
    ([4096] * 8, [2048] * 8, 16, 0.1, 0.0, "uniform"),                        # REF-like: 16 384 queries, S = 8
    ([900, 2500, 1300, 4000] * 9, [800, 2100, 1200, 3000] * 9, 8, 0.09, 0.02, "lattice"),  # 63 900 queries, S = 8, ragged, exact ties, blocks straddle clouds
    ([1024] * 80, [900] * 80, 16, 0.08, 0.0, "dup"),                          # 72 000 queries -> S = 4, duplicated points
    ([600] * 300, [500] * 300, 5, 0.15, 0.0, "uniform"),                      # 150 000 queries -> S = 2
    ([2048] * 20, [1700] * 20, 32, 0.25, 0.0, "uniform"),                     # most queries beyond the replay's 96 slots: flagged, redone wave-per-query
    ([300, 3000, 300], [300, 8000, 300], 16, 0.3, 0.0, "uniform"),            # one dense cloud: segment caps overflow for some queries only
]


@pytest.mark.parametrize("sizes,ms,nsample,rmax,rmin,mode", BALL_SPLIT)
def test_ball_query_split_path_bit_exact(hip_device, sizes, ms, nsample, rmax, rmin, mode):
    """From 8192 queries on `ball_query` runs as candidate collection (scalar-cache point reads, appends to a workspace) + heap replay
    out of LDS columns + the wave-per-query kernel for flagged queries: same bits as the oracle and as the single-kernel entry point."""
    from pointcloudmatters_amd import _lib
    from pointcloudmatters_amd.pointops.query import ball_query_raw
    from oracle import lib as olib

    xyz, off = make_clouds(sizes, seed=19, mode=mode)
    noff = new_offsets(ms)
    m = sum(ms)
    assert m >= 8192 and _lib.load().pcm_ball_query_ws_bytes(m) > 0
    starts = [0] + off.tolist()[:-1]
    sel = torch.cat([st + (torch.arange(mq) * 7919) % n for st, n, mq in zip(starts, sizes, ms)])
    new_xyz = xyz[sel].contiguous()
    wi = torch.zeros(m, nsample, dtype=torch.int32)
    wd = torch.zeros(m, nsample, dtype=torch.float32)
    rc = olib.load().pcm_ball_query_cpu(m, nsample, rmin, rmax, xyz.data_ptr(), new_xyz.data_ptr(), off.data_ptr(), noff.data_ptr(),
                                        wi.data_ptr(), wd.data_ptr())
    assert rc in (0, 2)
    d = hip_device
    gx, go, gq, gno = xyz.to(d), off.to(d), new_xyz.to(d), noff.to(d)
    gi, gd = ball_query_raw(nsample, rmax, rmin, gx, go, gq, gno)
    assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)
    # the reference-ABI entry point (no workspace: one kernel) gives the same rows
    L = _lib.load()
    bi, bd = torch.empty_like(gi), torch.empty_like(gd)
    assert L.pcm_ball_query_b_hip(len(ms), m, nsample, rmin, rmax, gx.data_ptr(), gq.data_ptr(), go.data_ptr(), gno.data_ptr(), bi.data_ptr(),
                                  bd.data_ptr(), _lib.raw_stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(bi, gi) and torch.equal(bd, gd)


@pytest.mark.parametrize("sizes,ms,nsample,rmax,rmin", [([1024] * 4, [256] * 4, 16, 0.1, 0.0), ([300, 20], [50, 20], 8, 0.3, 0.05)])
def test_random_ball_query_bit_exact(hip_device, sizes, ms, nsample, rmax, rmin):
    from pointcloudmatters_amd.pointops.query import random_ball_query_raw
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(sizes, seed=9)
    noff = new_offsets(ms)
    sel = ref.farthest_point_sampling(xyz, off, noff).long()
    new_xyz = xyz[sel].contiguous()
    g = torch.Generator().manual_seed(123)
    order = ref.make_random_order(off, generator=g)
    wi, wd = ref.random_ball_query_raw(nsample, rmax, rmin, xyz, off, new_xyz, noff, order)
    gi, gd = random_ball_query_raw(nsample, rmax, rmin, xyz.to(hip_device), off.to(hip_device), new_xyz.to(hip_device),
                                   noff.to(hip_device), order.to(hip_device))
    assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)


def _sa_inputs(sizes, ms, c, nsample, seed, mode="uniform"):
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(sizes, seed=seed, mode=mode)
    noff = new_offsets(ms)
    sel = ref.farthest_point_sampling(xyz, off, noff).long()
    new_xyz = xyz[sel].contiguous()
    idx, _ = ref.knn_query(nsample, xyz, off, new_xyz, noff)
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(xyz.shape[0], c, generator=g)
    return xyz, off, new_xyz, noff, idx, feat


@pytest.mark.parametrize("sizes,ms,c,with_xyz", [
    ([1024] * 4, [512] * 4, 512, True),
    ([1024] * 2, [512] * 2, 96, True),
    ([5, 3, 40], [5, 3, 20], 7, True),      # -1 placeholders (clouds smaller than nsample)
    ([5, 3, 40], [5, 3, 20], 8, False),
    ([600], [100], 64, False),
])
def test_grouping_forward_backward(hip_device, sizes, ms, c, with_xyz):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off, new_xyz, noff, idx, feat = _sa_inputs(sizes, ms, c, 16, seed=21)
    f_ref = feat.clone().requires_grad_(True)
    want = ref.grouping(idx, f_ref, xyz, new_xyz, with_xyz=with_xyz)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(gout)

    f_hip = feat.to(hip_device).requires_grad_(True)
    got = po.grouping(idx.to(hip_device), f_hip, xyz.to(hip_device), new_xyz.to(hip_device), with_xyz=with_xyz)
    assert torch.equal(got.detach().cpu(), want.detach())  # gather + one fp32 subtraction: exact
    got.backward(gout.to(hip_device))
    torch.testing.assert_close(f_hip.grad.cpu(), f_ref.grad, rtol=1e-5, atol=1e-5)


def test_grouping_xyz_gradients(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off, new_xyz, noff, idx, feat = _sa_inputs([5, 60], [5, 30], 4, 16, seed=4)
    a = [t.clone().requires_grad_(True) for t in (feat, xyz, new_xyz)]
    want = ref.grouping(idx, a[0], a[1], a[2], with_xyz=True)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
    want.backward(gout)
    b = [t.to(hip_device).requires_grad_(True) for t in (feat, xyz, new_xyz)]
    got = po.grouping(idx.to(hip_device), b[0], b[1], b[2], with_xyz=True)
    got.backward(gout.to(hip_device))
    for x, y in zip(a, b):
        torch.testing.assert_close(y.grad.cpu(), x.grad, rtol=1e-5, atol=1e-5)


def test_grouping2(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    for c in (512, 96, 5):
        xyz, off, new_xyz, noff, idx, feat = _sa_inputs([1024, 900], [256, 256], c, 16, seed=c)
        f_ref = feat.clone().requires_grad_(True)
        want = ref.grouping2(f_ref, idx)
        gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
        want.backward(gout)
        f_hip = feat.to(hip_device).requires_grad_(True)
        got = po.grouping2(f_hip, idx.to(hip_device))
        assert torch.equal(got.detach().cpu(), want.detach())
        got.backward(gout.to(hip_device))
        torch.testing.assert_close(f_hip.grad.cpu(), f_ref.grad, rtol=1e-5, atol=1e-5)


def test_knn_query_and_group(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off, new_xyz, noff, idx, feat = _sa_inputs([1024] * 2, [512] * 2, 32, 16, seed=8)
    want, widx = ref.knn_query_and_group(feat, xyz, offset=off, new_xyz=new_xyz, new_offset=noff, nsample=16, with_xyz=True)
    d = hip_device
    got, gidx = po.knn_query_and_group(feat.to(d), xyz.to(d), offset=off.to(d), new_xyz=new_xyz.to(d), new_offset=noff.to(d),
                                       nsample=16, with_xyz=True)
    assert torch.equal(gidx.cpu(), widx) and torch.equal(got.cpu(), want)
