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
