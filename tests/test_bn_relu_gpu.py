"""csrc/bnrelu.hip (fused BatchNorm1d + ReLU over (n, C) point features) against nn.BatchNorm1d + ReLU in fp32."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(c, seed):
    torch.manual_seed(seed)
    a = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(DEV)
    with torch.no_grad():
        a.weight.uniform_(-1.0, 1.5)  # negative scales too
        a.bias.uniform_(-0.5, 0.5)
        a.running_mean.normal_(0, 0.2)
        a.running_var.uniform_(0.5, 2.0)
    b = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(DEV)
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize("n,c", [(1, 64), (7, 64), (4096, 64), (8192, 128), (10000, 512), (333, 24), (2048, 1024), (300, 2048),
                                 (131072, 64)])
def test_bn_relu_matches_torch_fp32(n, c):
    from pointcloudmatters_amd.policy import bn_relu as fused

    ours, ref = _pair(c, n + c)
    y = (torch.randn(n, c, device=DEV) * 1.7 + 0.3).requires_grad_(True)
    assert fused.supported(y, ours)
    if n == 1:
        pytest.skip("BatchNorm1d rejects a single value per channel in training mode")
    z = fused.bn_relu(y, ours)
    want = torch.relu(ref(y))
    torch.testing.assert_close(z, want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ours.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ours.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(ours.num_batches_tracked) == 1
    g = torch.randn_like(want)
    got = torch.autograd.grad(z, (y, ours.weight, ours.bias), g)
    exp = torch.autograd.grad(want, (y, ref.weight, ref.bias), g)
    for a, r, name in zip(got, exp, ("dy", "dgamma", "dbeta")):
        scale = r.abs().max().item() + 1e-12
        assert (a - r).abs().max().item() <= 1e-4 * scale + 1e-6, (name, (a - r).abs().max().item(), scale)


def test_bn_relu_bf16_and_determinism():
    from pointcloudmatters_amd.policy import bn_relu as fused

    ours, ref = _pair(128, 3)
    y = torch.randn(5000, 128, device=DEV).bfloat16().requires_grad_(True)
    z = fused.bn_relu(y, ours)
    want = torch.relu(ref(y))
    assert z.dtype == torch.bfloat16 and want.dtype == torch.bfloat16
    torch.testing.assert_close(z.float(), want.float(), rtol=1e-2, atol=1e-2)
    g = torch.randn_like(want)
    a1 = torch.autograd.grad(z, (y, ours.weight, ours.bias), g, retain_graph=True)
    a2 = torch.autograd.grad(z, (y, ours.weight, ours.bias), g)
    exp = torch.autograd.grad(want, (y, ref.weight, ref.bias), g)
    for a, b, r in zip(a1, a2, exp):
        assert torch.equal(a, b)
        torch.testing.assert_close(a.float(), r.float(), rtol=3e-2, atol=3e-2)


def test_bn_relu_eval_uses_running_statistics():
    from pointcloudmatters_amd.policy import bn_relu as fused

    ours, ref = _pair(64, 9)
    ours.eval(), ref.eval()
    y = torch.randn(777, 64, device=DEV)
    with torch.no_grad():
        z = fused.bn_relu(y, ours)
        want = torch.relu(ref(y))
    torch.testing.assert_close(z, want, rtol=1e-5, atol=1e-6)
    assert int(ours.num_batches_tracked) == 0
    yg = y.clone().requires_grad_(True)  # gradients requested in eval mode: framework path, still correct
    fused.bn_relu(yg, ours).sum().backward()
    assert yg.grad is not None


def test_pointnet_uses_the_fused_tail_and_matches_cpu():
    from pointcloudmatters_amd.policy import PointNet

    torch.manual_seed(0)
    net = PointNet(in_channels=6, num_classes=96)
    feat = torch.randn(3000, 6)
    want = net({"feat": feat})
    want.square().mean().backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    ref_rm = net.conv5[1].running_mean.clone()
    net.zero_grad()
    net2 = PointNet(in_channels=6, num_classes=96)
    net2.load_state_dict({k: v for k, v in net.state_dict().items()})
    # rewind the running statistics the CPU pass advanced
    for m in net2.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.reset_running_stats()
    net2 = net2.to(DEV)
    got = net2({"feat": feat.to(DEV)})
    got.square().mean().backward()
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(net2.conv5[1].running_mean.cpu(), ref_rm, rtol=1e-4, atol=1e-6)
    for k, p in net2.named_parameters():
        r = ref[k]
        assert (p.grad.cpu() - r).abs().max() <= 2e-4 * r.abs().max() + 1e-7, k


def test_bn_relu_large_mean_small_spread_is_stable():
    """|mean| >> std: sum / sum-of-squares statistics would cancel; the kernel shifts by the first row."""
    from pointcloudmatters_amd.policy import bn_relu as fused

    ours, ref = _pair(64, 5)
    torch.manual_seed(0)
    y = (100.0 + 0.05 * torch.randn(4096, 64, device=DEV)).requires_grad_(True)
    z = fused.bn_relu(y, ours)
    want = torch.relu(ref(y))
    torch.testing.assert_close(z, want, rtol=2e-3, atol=2e-3)  # fp32 inputs at 1e2 with spread 5e-2: 1e-3-level conditioning
    torch.testing.assert_close(ours.running_var, ref.running_var, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("n,c", [(7, 64), (4096, 128), (128, 96), (333, 24)])
def test_bn_without_relu_matches_torch_fp32(n, c):
    """relu=False (csrc/bnrelu.hip pcm_bn_act_*): BatchNorm1d alone -- the last layer of the Diffusion Policy's projector."""
    from pointcloudmatters_amd.policy import bn_relu as fused

    ours, ref = _pair(c, n + c)
    y = (torch.randn(n, c, device=DEV) * 1.7 + 0.3).requires_grad_(True)
    assert fused.supported(y, ours)
    z = fused.bn_relu(y, ours, relu=False)
    want = ref(y)
    assert bool((z < 0).any())  # no clamp
    torch.testing.assert_close(z, want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ours.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ours.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    g = torch.randn_like(want)
    got = torch.autograd.grad(z, (y, ours.weight, ours.bias), g)
    exp = torch.autograd.grad(want, (y, ref.weight, ref.bias), g)
    for a, r, name in zip(got, exp, ("dy", "dgamma", "dbeta")):
        scale = r.abs().max().item() + 1e-12
        assert (a - r).abs().max().item() <= 1e-4 * scale + 1e-6, (name, (a - r).abs().max().item(), scale)
    ours.eval(), ref.eval()
    with torch.no_grad():
        torch.testing.assert_close(fused.bn_relu(y.detach(), ours, relu=False), ref(y.detach()), rtol=1e-4, atol=1e-5)


def test_diffusion_policy_projector_in_row_layout_equals_the_module_path(monkeypatch):
    """PCDObsEncoder's projector through csrc/bnrelu.hip in row layout (round 5) against the nn.Sequential it replaces: same features,
    same gradients; every BatchNorm of the policy is then owned by a fused kernel (what a captured data-parallel step needs)."""
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd.bc import BCTrainer, build_dp_policy, make_dp_batch
    from pointcloudmatters_amd.policy import diffusion
    from tests.golden.make_golden import DP_SMALL

    batch = make_dp_batch(3, 200, seed=12, ragged=True, device=DEV)
    runs, called = {}, []
    from pointcloudmatters_amd import _lib

    orig = _lib.check

    def check(rc, what, *a, **kw):
        called.append(what)
        return orig(rc, what, *a, **kw)

    monkeypatch.setattr(_lib, "check", check)
    for rows in (False, True):
        monkeypatch.setattr(diffusion, "PROJECTOR_ROWS", rows)
        torch.manual_seed(0)
        pol = build_dp_policy(pcd_npoints=32, pointops=po, sa_impl="fused", **DP_SMALL).to(DEV).train()
        # the projector's BatchNorms are claimed by the fused kernels only while the row-layout path is on (otherwise they stay with the
        # framework, which converts them to SyncBatchNorm under data parallelism: round-5 ADVICE)
        assert BCTrainer.all_batchnorms_fused(pol) == rows
        called.clear()
        enc = pol.obs_encoder
        feats = enc.pcd_features({k: v.clone() for k, v in batch["obs"]["pcds"].items()})
        n_bn = called.count("pcm_bn_relu_forward_hip")
        (feats * torch.linspace(-1, 1, feats.numel(), device=DEV).view_as(feats)).sum().backward()
        runs[rows] = (feats.detach(), {n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None},
                      {n: b.detach().clone() for n, b in enc.named_buffers() if "running" in n}, n_bn)
    assert runs[True][3] == runs[False][3] + 2  # the projector's two BatchNorms joined the PointNet's and the SA layer's
    torch.testing.assert_close(runs[True][0], runs[False][0], rtol=1e-4, atol=1e-5)
    assert runs[True][1].keys() == runs[False][1].keys()
    gscale = max(r.abs().max().item() for r in runs[False][1].values())
    for n in runs[False][1]:  # (convolution biases in front of a BatchNorm have a true gradient of zero: rounding noise on both sides)
        a, r = runs[True][1][n], runs[False][1][n]
        assert (a - r).abs().max().item() <= 1e-4 * max(r.abs().max().item(), 1e-2 * gscale) + 1e-6, n
    for n in runs[False][2]:
        torch.testing.assert_close(runs[True][2][n], runs[False][2][n], rtol=1e-4, atol=1e-6, msg=n)
