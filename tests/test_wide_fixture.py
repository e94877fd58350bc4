"""The fused HIP kernels against the REFERENCE at the shipped widths.

Fixture tests/golden/wide_ref.npz (tests/golden/make_golden.py ``wide``): the reference's own ACTPCD / Transformer /
KLDivergence at d = 512, 8 heads, feed-forward 32, 100 queries (configs/model/maniskill2_act_pcd_model.yaml:49-68) with
1 encoder + 2 decoder layers on two ragged ~150-point clouds -- variant "flash" with 128 tokens per cloud (131-token
sequences: csrc/attn_flash.hip), variant "small" with 96 (99 tokens: csrc/attn_small.hip) -- and the reference's
PCDObsEncoder + ConditionalUnet1D at the shipped encoder widths (PointNet head 96, SA 96, projector [96, 128, 128]).  Weights
come from tests/util.seeded_fill on both sides; the fixture stores inputs, outputs and a digest of EVERY gradient.

These are the widths at which the fused kernels engage (E % 256 == 0, head_dim 64, feed-forward 512 / 32): the small
fixtures (hidden 48) never reach them.  fp32 with the fused context: 1e-4 relative (north_star); bf16: per-tensor bounds.
"""
import contextlib

import numpy as np
import pytest
import torch

from tests.test_golden_cpu import ATOL, RTOL, _load
from tests.util import check_grad_digest, seeded_fill

ACT_TAGS = [("flash", 128), ("small", 96)]


def _digests(fx, prefix):
    out = {}
    for k in fx.files:
        if k.startswith(prefix):
            name, part = k[len(prefix):].rsplit("/", 1)
            out.setdefault(name, {})[part] = fx[k]
    return out


def _act_case(tag, M, pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_act_policy
    from tests.golden.make_golden import WIDE, WIDE_SEED

    fx = _load("wide_ref.npz")
    pol = build_act_policy(pcd_npoints=M, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", **WIDE)
    assert seeded_fill(pol, WIDE_SEED) == float(fx[f"act.{tag}.wsum"])
    pol = pol.to(device).train()
    pre = f"act.{tag}.in."
    batch = {"pcds": {}}
    for k in fx.files:
        if k.startswith(pre + "pcds."):
            batch["pcds"][k[len(pre) + 5:]] = torch.from_numpy(fx[k]).to(device)
        elif k.startswith(pre):
            batch[k[len(pre):]] = torch.from_numpy(fx[k]).to(device)
    batch["vae_eps"] = torch.from_numpy(fx["act.eps"]).to(device)
    return fx, pol, batch


@contextlib.contextmanager
def _recorded_launches():
    """Names of the C-ABI entry points called inside the block (every call site reports through _lib.check)."""
    from pointcloudmatters_amd import _lib

    seen, orig = [], _lib.check

    def check(rc, what, *a, **kw):
        seen.append(what)
        return orig(rc, what, *a, **kw)

    _lib.check = check
    try:
        yield seen
    finally:
        _lib.check = orig


def _run_act(pol, batch, fused, bf16):
    from pointcloudmatters_amd.policy import fused_ops

    ctx = fused_ops.FusedContext(batch["qpos"].device) if fused else None
    ac = torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else contextlib.nullcontext()
    with fused_ops.activate(ctx), ac:
        out = pol(batch)
    out["loss"].backward()
    return out


def _check_act_outputs(fx, tag, pol, out, rtol, atol, grad_rtol):
    for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src", "pos"):
        ref = fx[f"act.{tag}.out.{k}"]
        got = out[k].detach().float().cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=k)
    grads = dict(pol.named_parameters())
    dig = _digests(fx, f"act.{tag}.grad.")
    assert len(dig) >= 90  # every parameter with a gradient: PointNet, SA, CVAE encoder, encoder, both decoder layers, heads
    worst = {}
    for name, ref in dig.items():
        worst[name] = check_grad_digest(name, grads[name].grad.detach().float().cpu().numpy(), ref, rtol=grad_rtol)
    assert set(fx[f"act.{tag}.grad_none"].tolist()) == {n for n, p in pol.named_parameters() if p.grad is None}
    return worst


@pytest.mark.parametrize("tag,M", ACT_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_act_wide_matches_reference_cpu(sa_impl, tag, M):
    from oracle import pointops_cpu

    fx, pol, batch = _act_case(tag, M, pointops_cpu, sa_impl)
    out = pol(batch)
    out["loss"].backward()
    _check_act_outputs(fx, tag, pol, out, RTOL, ATOL, 1e-4)
    np.testing.assert_allclose(pol.bn.running_mean.numpy(), fx[f"act.{tag}.bn_running_mean"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pol.bn.running_var.numpy(), fx[f"act.{tag}.bn_running_var"], rtol=RTOL, atol=ATOL)


def _dp_case(pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_dp_policy
    from tests.golden.make_golden import WIDE_DP, WIDE_SEED

    fx = _load("wide_ref.npz")
    pol = build_dp_policy(pcd_npoints=64, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", **WIDE_DP)
    assert seeded_fill(pol, WIDE_SEED + 1) == float(fx["dp.wsum"])
    pol = pol.to(device).train()
    pcds = {k[len("dp.in.pcds."):]: torch.from_numpy(fx[k]).to(device) for k in fx.files if k.startswith("dp.in.pcds.")}
    batch = {"obs": {"pcds": pcds, "qpos": torch.from_numpy(fx["dp.in.qpos"]).to(device)},
             "action": torch.from_numpy(fx["dp.in.action"]).to(device), "noise": torch.from_numpy(fx["dp.noise"]).to(device),
             "timesteps": torch.from_numpy(fx["dp.timesteps"]).to(device)}
    return fx, pol, batch


def _check_dp(fx, pol, out, rtol, grad_rtol):
    np.testing.assert_allclose(out["loss"].detach().float().cpu().numpy(), fx["dp.out.loss"], rtol=rtol, atol=ATOL)
    grads = dict(pol.named_parameters())
    dig = _digests(fx, "dp.grad.")
    assert len(dig) >= 100
    for name, ref in dig.items():
        check_grad_digest(name, grads[name].grad.detach().float().cpu().numpy(), ref, rtol=grad_rtol)


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_dp_wide_matches_reference_cpu(sa_impl):
    from oracle import pointops_cpu

    fx, pol, batch = _dp_case(pointops_cpu, sa_impl)
    out = pol(batch)
    out["loss"].backward()
    _check_dp(fx, pol, out, RTOL, 1e-4)


# ----------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,M", ACT_TAGS)
def test_act_wide_fp32_fused_matches_reference_gpu(hip_device, tag, M):
    """fp32, fused context active: drln / proj_drln, ffn_ln, the fused SA layer, bn_relu, the ACT loss and the CVAE latent
    kernels against the reference's loss, a_hat, mu / logvar and every gradient at 1e-4."""
    import pointcloudmatters_amd.pointops as po

    fx, pol, batch = _act_case(tag, M, po, "fused", device=hip_device)
    with _recorded_launches() as seen:
        out = _run_act(pol, batch, fused=True, bf16=False)
    names = set(seen)
    for must in ("pcm_sa_fused_forward_hip", "pcm_sa_fused_backward_hip", "pcm_act_loss_forward_hip", "pcm_cvae_latent_forward_hip"):
        assert must in names, (must, sorted(names))
    assert any(n.startswith("pcm_drln") or n.startswith("pcm_proj_drln") for n in names), sorted(names)
    assert any(n.startswith("pcm_ffn_ln") for n in names), sorted(names)
    assert any(n.startswith("pcm_bn_relu") for n in names), sorted(names)
    _check_act_outputs(fx, tag, pol, out, RTOL, ATOL, 1e-4)
    np.testing.assert_allclose(pol.bn.running_mean.cpu().numpy(), fx[f"act.{tag}.bn_running_mean"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pol.bn.running_var.cpu().numpy(), fx[f"act.{tag}.bn_running_var"], rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,M", ACT_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "fused"])
def test_act_wide_fp32_eager_matches_reference_gpu(hip_device, tag, M, sa_impl):
    """The same comparison without the fused transformer tails (what mode="eager" runs)."""
    import pointcloudmatters_amd.pointops as po

    fx, pol, batch = _act_case(tag, M, po, sa_impl, device=hip_device)
    out = _run_act(pol, batch, fused=False, bf16=False)
    _check_act_outputs(fx, tag, pol, out, RTOL, ATOL, 1e-4)


# ---- bf16: the configuration bench.py times, against the SAME fp32 reference numbers, tensor by tensor.
# What bf16 can and cannot be held to was measured first (tools/dbg/wide_bf16_errors.py on MI355X): the median tensor is 0.6-1 %
# from the reference (relative to its largest magnitude), but two groups are far out in ANY bf16 evaluation, the framework's
# included, because a bf16-sized perturbation (1e-2) flips ReLU gates / arg-max choices that own a large share of the gradient:
#   * the tokenizer (PointNet behind five training-mode BatchNorms on ~300 rows, the SA layer's arg-max): 8-27 %;
#   * the CVAE encoder: only its CLS token reaches the loss, so 2 x 32 feed-forward gates decide `linear1`'s whole gradient
#     (one flipped gate = one row of the matrix): up to 45 %.
# So each tensor of the fused bf16 path must be (a) within its class cap -- which a wrong sign (error 2), a missing gradient
# (error 1) or a dropped term still break -- and (b) outside the flip-prone classes no further from the reference than three times
# what PyTorch's own bf16 autocast of the reference op order is on the same tensor (floor 5 %).
BF16_OUT_RTOL = 3e-2


def _bf16_class(name):
    if name.startswith(("backbone.", "obs_encoder.")) or name in ("linear.weight", "bn.weight", "bn.bias"):
        return "tokenizer", 0.8
    if name.startswith("encoder.") or name in ("encoder_action_proj.weight", "encoder_action_proj.bias", "encoder_joint_proj.weight",
                                               "encoder_joint_proj.bias", "cls_embed.weight", "latent_proj.weight", "latent_proj.bias"):
        return "cvae_encoder", 0.8
    return "rest", 0.2


def _tensor_errors(fx, prefix, pol):
    from tests.util import digest_rel_error

    grads = dict(pol.named_parameters())
    return {name: digest_rel_error(name, grads[name].grad.detach().float().cpu().numpy(), ref) for name, ref in _digests(fx, prefix).items()}


def _judge_bf16(err_fused, err_eager, floor_rest=0.05, cap_rest=0.2):
    bad = []
    for name, (e, scale) in err_fused.items():
        if scale < 1e-6:  # numerically-zero gradients (the decoder's first self-attention acts on an all-zero target)
            continue
        cls, cap = _bf16_class(name)
        if cls == "rest":
            cap = cap_rest
        bound = cap if cls != "rest" else min(cap, max(3.0 * err_eager[name][0], floor_rest))
        if e > bound:
            bad.append((name, cls, round(e, 4), round(err_eager[name][0], 4), round(bound, 4)))
    assert not bad, bad
    rest = [n for n in err_fused if _bf16_class(n)[0] == "rest" and err_fused[n][1] >= 1e-6]
    med_f, med_e = np.median([err_fused[n][0] for n in rest]), np.median([err_eager[n][0] for n in rest])
    assert med_f <= 2.0 * med_e + 5e-3, (med_f, med_e)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,M", ACT_TAGS)
def test_act_wide_bf16_fused_matches_reference_gpu(hip_device, tag, M):
    import pointcloudmatters_amd.pointops as po

    fx, pol, batch = _act_case(tag, M, po, "fused", device=hip_device)
    with _recorded_launches() as seen:
        out = _run_act(pol, batch, fused=True, bf16=True)
    names = set(seen)
    want_attn = "pcm_attn_flash_forward_hip" if tag == "flash" else "pcm_attn_small_forward_hip"
    assert want_attn in names and "pcm_attn_small_forward_hip" in names, sorted(names)  # decoder / CVAE encoder: short query sets
    assert any(n.startswith("pcm_ffn_ln") for n in names) and any("drln" in n for n in names), sorted(names)
    for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src"):
        ref = fx[f"act.{tag}.out.{k}"]
        got = out[k].detach().float().cpu().numpy()
        assert np.abs(got - ref).max() <= BF16_OUT_RTOL * np.abs(ref).max(), k
    err_fused = _tensor_errors(fx, f"act.{tag}.grad.", pol)
    # the yardstick: PyTorch's own bf16 kernels in the reference op order (no fused context, reference SA layer)
    _, pol_e, batch_e = _act_case(tag, M, po, "reference", device=hip_device)
    _run_act(pol_e, batch_e, fused=False, bf16=True)
    _judge_bf16(err_fused, _tensor_errors(fx, f"act.{tag}.grad.", pol_e))


@pytest.mark.gpu
@pytest.mark.parametrize("sa_impl", ["reference", "torch", "fused"])
def test_dp_wide_fp32_matches_reference_gpu(hip_device, sa_impl):
    import pointcloudmatters_amd.pointops as po

    fx, pol, batch = _dp_case(po, sa_impl, device=hip_device)
    with _recorded_launches() as seen:
        out = pol(batch)
        out["loss"].backward()
    names = set(seen)
    assert "pcm_gn_mish_forward_hip" in names and "pcm_gn_mish_backward_hip" in names and "pcm_im2col_cl_hip" in names, sorted(names)
    if sa_impl == "fused":
        assert "pcm_sa_fused_forward_hip" in names
    _check_dp(fx, pol, out, RTOL, 1e-4)


@pytest.mark.gpu
def test_dp_wide_bf16_matches_reference_gpu(hip_device):
    """Diffusion Policy in bf16: two samples whose 96 projector channels each pick ONE token (MaxPool over the cloud) make every
    encoder gradient flip-prone, and the U-Net sees the perturbed condition: measured 12 % median, the framework's autocast the
    same.  Judged like the ACT case, with the U-Net as the "rest" class at a 10 % floor / 60 % cap."""
    import pointcloudmatters_amd.pointops as po

    def run(sa_impl):
        fx, pol, batch = _dp_case(po, sa_impl, device=hip_device)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = pol(batch)
        out["loss"].backward()
        return fx, pol, out

    fx, pol, out = run("fused")
    np.testing.assert_allclose(out["loss"].detach().float().cpu().numpy(), fx["dp.out.loss"], rtol=BF16_OUT_RTOL)
    _, pol_e, _ = run("reference")
    _judge_bf16(_tensor_errors(fx, "dp.grad.", pol), _tensor_errors(fx, "dp.grad.", pol_e), floor_rest=0.10, cap_rest=0.6)
