"""GPU parity for the API-only pointops (K6-K9: interpolation, subtraction, aggregation, attention
steps) and the *_and_group helpers: HIP through the C ABI vs the CPU oracle.  Forward results that
are pure gathers / fixed-order sums are exact; atomic-scatter results within 1e-5 relative."""
import pytest
import torch

from tests.util import make_clouds

pytestmark = pytest.mark.gpu


def _both(fn_ref, fn_hip, inputs, dev, grad_idx, exact_fwd):
    a = [t.clone().requires_grad_(True) if i in grad_idx else t for i, t in enumerate(inputs)]
    b = [t.to(dev).clone().requires_grad_(True) if i in grad_idx else t.to(dev) for i, t in enumerate(inputs)]
    out_ref = fn_ref(*a)
    out_hip = fn_hip(*b)
    if exact_fwd:
        assert torch.equal(out_hip.detach().cpu(), out_ref.detach())
    else:
        torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), rtol=1e-5, atol=1e-5)
    gout = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(0))
    out_ref.backward(gout)
    out_hip.backward(gout.to(dev))
    for i in grad_idx:
        torch.testing.assert_close(b[i].grad.cpu(), a[i].grad, rtol=1e-4, atol=1e-5)


def test_interpolation(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds([300, 200], seed=1)
    new_xyz, noff = make_clouds([500, 100], seed=2)
    feat = torch.randn(500, 32, generator=torch.Generator().manual_seed(3))
    for k in (3, 5):
        _both(lambda a, b, c, d, e: ref.interpolation2(a, b, c, d, e, k), lambda a, b, c, d, e: po.interpolation2(a, b, c, d, e, k),
              [xyz, new_xyz, feat, off, noff], hip_device, grad_idx=[2], exact_fwd=False)
        got = po.interpolation(xyz.to(hip_device), new_xyz.to(hip_device), feat.to(hip_device), off.to(hip_device), noff.to(hip_device), k)
        want = ref.interpolation(xyz, new_xyz, feat, off, noff, k)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)


def test_subtraction(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(4)
    n, k, c = 700, 16, 48
    a, b = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    idx = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
    _both(ref.subtraction, po.subtraction, [a, b, idx], hip_device, grad_idx=[0, 1], exact_fwd=True)


def test_aggregation(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(5)
    n, k, c, wc = 400, 8, 32, 8
    inp, pos, w = torch.randn(n, c, generator=g), torch.randn(n, k, c, generator=g), torch.randn(n, k, wc, generator=g)
    idx = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
    _both(ref.aggregation, po.aggregation, [inp, pos, w, idx], hip_device, grad_idx=[0, 1, 2], exact_fwd=True)


def test_attention_steps(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(6)
    n, m, gg, c = 300, 2000, 4, 16
    q, k = torch.randn(n, gg, c, generator=g), torch.randn(n, gg, c, generator=g)
    w = torch.ones(c)
    it = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    ir = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    _both(ref.attention_relation_step, po.attention_relation_step, [q, k, w, it, ir], hip_device, grad_idx=[0, 1], exact_fwd=True)
    aw, v = torch.randn(m, gg, generator=g), torch.randn(n, gg, c, generator=g)
    _both(ref.attention_fusion_step, po.attention_fusion_step, [aw, v, it, ir], hip_device, grad_idx=[0, 1], exact_fwd=False)


# ---------------------------------------------------------------------------------------------------------------------
# Widened coverage (round 4): config-size widths (c = 96 / 512), n >= 32 768, widths that are not a multiple of 4 (the
# kernels' 16-byte paths must fall back), k in {1, 3, 5}, an empty cloud in the middle of the batch, hub rows (one source row
# referenced by thousands of entries) -- every case against the C oracle, which restates
# libs/pointops/src/{interpolation,subtraction,aggregation,attention}/*_cuda_kernel.cu.  The reference kernels index with
# idx directly (no -1 handling: a placeholder there is undefined behaviour), so -1 lists are exercised where the reference
# tolerates them: grouping() / the *_and_group helpers below.
def _both_scaled(fn_ref, fn_hip, inputs, dev, grad_idx, exact_fwd, seed=0):
    a = [t.clone().requires_grad_(True) if i in grad_idx else t for i, t in enumerate(inputs)]
    b = [t.to(dev).clone().requires_grad_(True) if i in grad_idx else t.to(dev) for i, t in enumerate(inputs)]
    out_ref, out_hip = fn_ref(*a), fn_hip(*b)
    assert out_hip.shape == out_ref.shape and out_hip.dtype == out_ref.dtype
    if exact_fwd:
        assert torch.equal(out_hip.detach().cpu(), out_ref.detach())
    else:
        scale = float(out_ref.detach().abs().max()) if out_ref.numel() else 1.0
        torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), rtol=1e-5, atol=1e-5 * max(scale, 1.0))
    if out_ref.numel() == 0:
        return
    gout = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(seed))
    out_ref.backward(gout)
    out_hip.backward(gout.to(dev))
    for i in grad_idx:
        scale = float(a[i].grad.abs().max())
        torch.testing.assert_close(b[i].grad.cpu(), a[i].grad, rtol=1e-4, atol=1e-5 * max(scale, 1.0))


@pytest.mark.parametrize("src,dst,c,k", [
    ([300, 200], [500, 100], 32, 3),
    ([20000, 13000], [30000, 10000], 96, 3),       # n = 40 000 query rows, the DP width
    ([9000, 9000], [16500, 16500], 512, 3),        # n = 33 000, the ACT width
    ([700, 50, 400], [1000, 30, 900], 1, 1),       # one channel, nearest neighbour
    ([2000, 1500], [1800, 1700], 7, 5),            # c % 4 != 0, k = 5
    ([300, 0, 200], [500, 0, 100], 4, 3),          # an empty cloud in the middle of the batch
    ([64], [4096], 20, 3),                         # up-sampling from a tiny cloud: every source row is a hub
])
def test_interpolation_cases(hip_device, src, dst, c, k):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds(src, seed=11)
    new_xyz, noff = make_clouds(dst, seed=12)
    feat = torch.randn(sum(src), c, generator=torch.Generator().manual_seed(13))
    _both_scaled(lambda a, b, f, o, no: ref.interpolation2(a, b, f, o, no, k), lambda a, b, f, o, no: po.interpolation2(a, b, f, o, no, k),
                 [xyz, new_xyz, feat, off, noff], hip_device, grad_idx=[2], exact_fwd=False)
    d = hip_device
    got = po.interpolation(xyz.to(d), new_xyz.to(d), feat.to(d), off.to(d), noff.to(d), k)
    want = ref.interpolation(xyz, new_xyz, feat, off, noff, k)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,k,c", [(700, 16, 48), (33000, 16, 96), (2000, 1, 512), (1000, 3, 7), (500, 5, 1), (40000, 3, 12), (257, 16, 130)])
def test_subtraction_cases(hip_device, n, k, c):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(n + k + c)
    a, b = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    idx = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
    idx[: n // 3] = torch.randint(0, 3, (n // 3, k), generator=g, dtype=torch.int32)  # hub rows: three targets take a third of the entries
    _both_scaled(ref.subtraction, po.subtraction, [a, b, idx], hip_device, grad_idx=[0, 1], exact_fwd=True)


@pytest.mark.parametrize("n,k,c,wc", [(400, 8, 32, 8), (33000, 16, 96, 12), (1500, 3, 512, 64), (900, 5, 7, 1), (800, 1, 1, 1),
                                     (2000, 16, 96, 96), (3000, 5, 30, 6)])
def test_aggregation_cases(hip_device, n, k, c, wc):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(n + k + c + wc)
    inp, pos, w = torch.randn(n, c, generator=g), torch.randn(n, k, c, generator=g), torch.randn(n, k, wc, generator=g)
    idx = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
    idx[::4] = torch.randint(0, 2, (idx[::4].shape[0], k), generator=g, dtype=torch.int32)
    _both_scaled(ref.aggregation, po.aggregation, [inp, pos, w, idx], hip_device, grad_idx=[0, 1, 2], exact_fwd=True)


@pytest.mark.parametrize("n,m,gg,c", [(300, 2000, 4, 16), (33000, 200000, 8, 12), (1000, 5000, 1, 7), (500, 3000, 3, 1), (2000, 40000, 6, 16),
                                     (64, 100000, 2, 8)])
def test_attention_step_cases(hip_device, n, m, gg, c):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    g = torch.Generator().manual_seed(n + m)
    q, k = torch.randn(n, gg, c, generator=g), torch.randn(n, gg, c, generator=g)
    w = torch.ones(c)
    it = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    ir = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
    _both_scaled(ref.attention_relation_step, po.attention_relation_step, [q, k, w, it, ir], hip_device, grad_idx=[0, 1], exact_fwd=c <= 16)
    aw, v = torch.randn(m, gg, generator=g), torch.randn(n, gg, c, generator=g)
    _both_scaled(ref.attention_fusion_step, po.attention_fusion_step, [aw, v, it, ir], hip_device, grad_idx=[0, 1], exact_fwd=False)


@pytest.mark.parametrize("sizes,c,k,radius", [([400, 300], 8, 8, 0.08), ([5000, 3000], 96, 16, 0.02), ([900], 7, 5, 0.01), ([1200, 0, 800], 1, 16, 0.05),
                                             ([33000], 12, 3, 0.004), ([600, 700], 512, 16, 0.03)])
def test_grouping_with_placeholder_rows_from_a_ball_query(hip_device, sizes, c, k, radius):
    """Where the reference DOES tolerate -1 (functions/grouping.py:40-57: appended zero row + mask): neighbour lists of a ball
    query whose radius leaves most slots empty, through ball_query_and_group and through grouping() with its backward."""
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    d = hip_device
    xyz, off = make_clouds(sizes, seed=17)
    feat = torch.randn(sum(sizes), c, generator=torch.Generator().manual_seed(18))
    want, widx = ref.ball_query_and_group(feat, xyz, offset=off, max_radio=radius, min_radio=0.0, nsample=k, with_xyz=True)
    got, gidx = po.ball_query_and_group(feat.to(d), xyz.to(d), offset=off.to(d), max_radio=radius, min_radio=0.0, nsample=k, with_xyz=True)
    assert torch.equal(gidx.cpu(), widx) and torch.equal(got.cpu(), want)
    if k > 1:
        assert int((widx < 0).sum()) > 0, "the case is meant to contain placeholders"
    for with_xyz in (True, False):
        _both_scaled(lambda f: ref.grouping(widx, f, xyz, xyz, with_xyz=with_xyz),
                     lambda f: po.grouping(widx.to(d), f, xyz.to(d), xyz.to(d), with_xyz=with_xyz), [feat], d, grad_idx=[0], exact_fwd=True)


def test_misc_ops_seeded_sweep(hip_device):
    """The K6-K9 part of tools/fuzz_pointops.py inside the suite: 40 seeded random shapes per op (< 10 s)."""
    import numpy as np

    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    for seed in range(40):
        rng = np.random.default_rng(7000 + seed)
        g = torch.Generator().manual_seed(seed)
        n, k, c = int(rng.integers(1, 3000)), int(rng.choice([1, 2, 3, 5, 8, 16, 31])), int(rng.choice([1, 2, 3, 4, 7, 8, 12, 33, 64, 96]))
        a, b = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
        idx = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
        _both_scaled(ref.subtraction, po.subtraction, [a, b, idx], hip_device, grad_idx=[0, 1], exact_fwd=True, seed=seed)
        wc = int(rng.choice([w for w in (1, 2, 3, 4, c) if c % w == 0]))
        pos, w = torch.randn(n, k, c, generator=g), torch.randn(n, k, wc, generator=g)
        _both_scaled(ref.aggregation, po.aggregation, [a, pos, w, idx], hip_device, grad_idx=[0, 1, 2], exact_fwd=True, seed=seed)
        m, gg = int(rng.integers(1, 20000)), int(rng.choice([1, 2, 4, 6]))
        q, kk = torch.randn(n, gg, c, generator=g), torch.randn(n, gg, c, generator=g)
        it = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
        ir = torch.randint(0, n, (m,), generator=g, dtype=torch.int32)
        _both_scaled(ref.attention_relation_step, po.attention_relation_step, [q, kk, torch.ones(c), it, ir], hip_device, grad_idx=[0, 1],
                     exact_fwd=False, seed=seed)
        aw = torch.randn(m, gg, generator=g)
        _both_scaled(ref.attention_fusion_step, po.attention_fusion_step, [aw, q, it, ir], hip_device, grad_idx=[0, 1], exact_fwd=False, seed=seed)
        b_cl = int(rng.integers(1, 4))
        src = [int(rng.integers(k, 1500)) for _ in range(b_cl)]
        dst = [int(rng.integers(1, 2500)) for _ in range(b_cl)]
        xyz, off = make_clouds(src, seed=seed)
        new_xyz, noff = make_clouds(dst, seed=seed + 1)
        feat = torch.randn(sum(src), c, generator=g)
        ki = int(rng.choice([1, 3, 5]))
        _both_scaled(lambda x, y, f, o, no: ref.interpolation2(x, y, f, o, no, ki), lambda x, y, f, o, no: po.interpolation2(x, y, f, o, no, ki),
                     [xyz, new_xyz, feat, off, noff], hip_device, grad_idx=[2], exact_fwd=False, seed=seed)


def test_query_and_group_helpers(hip_device):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    d = hip_device
    xyz, off = make_clouds([400, 300], seed=7)
    feat = torch.randn(700, 8, generator=torch.Generator().manual_seed(8))
    want, widx = ref.ball_query_and_group(feat, xyz, offset=off, max_radio=0.08, min_radio=0.0, nsample=8, with_xyz=True)
    got, gidx = po.ball_query_and_group(feat.to(d), xyz.to(d), offset=off.to(d), max_radio=0.08, min_radio=0.0, nsample=8, with_xyz=True)
    assert torch.equal(gidx.cpu(), widx) and torch.equal(got.cpu(), want)
    want, widx = ref.query_and_group(8, xyz, xyz, feat, None, off, off, dilation=1)
    got, gidx = po.query_and_group(8, xyz.to(d), xyz.to(d), feat.to(d), None, off.to(d), off.to(d), dilation=1)
    assert torch.equal(gidx.cpu(), widx) and torch.equal(got.cpu(), want)
    assert torch.equal(po.offset2batch(off.to(d)).cpu(), ref.offset2batch(off))
    assert torch.equal(po.batch2offset(po.offset2batch(off.to(d))).cpu(), off)


def test_pybind_surface_with_the_reference_wrapper_call_pattern(hip_device):
    """pointops._C replacement: the 16 pybind names, called exactly like the reference wrappers do
    (functions/sampling.py:14-23, query.py:17-23, grouping.py:16-18): caller-allocated outputs, tmp pre-filled."""
    from oracle import pointops_cpu as ref
    from pointcloudmatters_amd.pointops import _C
    from tests.util import make_clouds, new_offsets

    for name in ("knn_query_cuda", "ball_query_cuda", "random_ball_query_cuda", "farthest_point_sampling_cuda",
                 "grouping_forward_cuda", "grouping_backward_cuda", "interpolation_forward_cuda", "interpolation_backward_cuda",
                 "subtraction_forward_cuda", "subtraction_backward_cuda", "aggregation_forward_cuda", "aggregation_backward_cuda",
                 "attention_relation_step_forward_cuda", "attention_relation_step_backward_cuda",
                 "attention_fusion_step_forward_cuda", "attention_fusion_step_backward_cuda"):
        assert callable(getattr(_C, name))  # pointops_api.cpp:16-31
    xyz_c, off_c = make_clouds([900, 500], seed=2)
    noff_c = new_offsets([256, 128])
    xyz, offset, new_offset = xyz_c.to(hip_device), off_c.to(hip_device), noff_c.to(hip_device)
    n, b, n_max = xyz.shape[0], offset.shape[0], 900
    idx = torch.zeros(384, dtype=torch.int32, device=hip_device)
    tmp = torch.full((n,), 1e10, dtype=torch.float32, device=hip_device)
    _C.farthest_point_sampling_cuda(b, n_max, xyz, offset.int(), new_offset.int(), tmp, idx)
    want = ref.farthest_point_sampling(xyz_c, off_c, noff_c)
    assert torch.equal(idx.cpu(), want)
    new_xyz = xyz[idx.long()].contiguous()
    kidx = torch.zeros(384, 16, dtype=torch.int32, device=hip_device)
    dist2 = torch.zeros(384, 16, dtype=torch.float32, device=hip_device)
    _C.knn_query_cuda(384, 16, xyz, new_xyz, offset.int(), new_offset.int(), kidx, dist2)
    wi, wd = ref.knn_query_raw(16, xyz_c, off_c, xyz_c[want.long()].contiguous(), noff_c)
    assert torch.equal(kidx.cpu(), wi) and torch.equal(dist2.cpu(), wd)
    feat = torch.randn(n, 24, device=hip_device)
    out = torch.empty(384, 16, 24, device=hip_device)
    _C.grouping_forward_cuda(384, 16, 24, feat, kidx, out)
    assert torch.equal(out, feat[kidx.long()])
    with pytest.raises(TypeError):
        _C.knn_query_cuda(384, 16, xyz, new_xyz, offset.long(), new_offset.int(), kidx, dist2)  # int64 offsets, like data_ptr<int>()
