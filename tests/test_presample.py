"""``pre_sample=True``: the set-abstraction layer in FRONT of the backbone (reference act.py:366-376,509-530;
pcd_obs_encoder.py:81-120,200-218) -- the branch configs/exp_maniskill2_{act,diffusion}_policy/maniskill2_model/
scratch_pointnet_pcd_presample{,_wo_rgb,_wo_xyz}.yaml select.

Fixture: tests/golden/presample_ref.npz -- the reference's ACTPCD / PCDObsEncoder (+ ConditionalUnet1D composed as
compute_loss does) run by tests/golden/make_golden.py ``presample`` with feature widths 6 and 3, with and without
``use_mask``.  Weights are regenerated on both sides by tests/util.seeded_fill (checksum stored).  Sampled indices exact;
tokens / losses / gradients 1e-4 relative (BASELINE.json north_star).
"""
import numpy as np
import pytest
import torch

from tests.test_golden_cpu import ATOL, RTOL, _load

ACT_TAGS = [("c6", 6, False), ("c3", 3, False), ("c6mask", 6, True)]
DP_TAGS = [("c6", 6), ("c3", 3)]


def _act_case(tag, cin, use_mask, pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_act_policy
    from tests.golden.make_golden import PRESAMPLE_SEED, SMALL
    from tests.util import seeded_fill

    fx = _load("presample_ref.npz")
    pol = build_act_policy(pcd_npoints=32, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", pre_sample=True,
                           in_channels=cin, backbone_num_classes=SMALL["hidden_dim"],
                           **({"use_mask": True, "bg_ratio": 0.25} if use_mask else {}), **SMALL)
    assert seeded_fill(pol, PRESAMPLE_SEED) == float(fx[f"act.{tag}.wsum"])
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


def _check_act(tag, cin, use_mask, pointops, sa_impl, device="cpu"):
    fx, pol, batch = _act_case(tag, cin, use_mask, pointops, sa_impl, device)
    assert tuple(pol.linear.weight.shape) == (cin, 3 + cin) and pol.bn.num_features == cin
    n_in = batch["pcds"]["coord"].shape[0]
    idx = pol.sampling_for(batch["pcds"], overlap=False)["idx"]
    assert np.array_equal(idx.cpu().numpy(), fx[f"act.{tag}.idx"])
    pcds = batch["pcds"]
    out = pol(batch)
    out["loss"].backward()
    # like the reference, the cloud dict now describes the SAMPLED cloud (act.py:524-527)
    assert pcds["coord"].shape[0] == 3 * 32 < n_in and pcds["feat"].shape == (96, cin) and pcds["grid_coord"].shape == (96, 3)
    assert pcds["offset"].tolist() == [32, 64, 96]
    for k in ("a_hat", "is_pad_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src", "pos"):
        np.testing.assert_allclose(out[k].detach().float().cpu().numpy(), fx[f"act.{tag}.out.{k}"], rtol=RTOL, atol=ATOL, err_msg=k)
    grads = dict(pol.named_parameters())
    gp = f"act.{tag}.grad."
    n = 0
    for k in fx.files:
        if k.startswith(gp):
            ref = fx[k]
            g = grads[k[len(gp):]].grad.detach().cpu().numpy()
            assert np.abs(g - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12) + 1e-6, k
            n += 1
    assert n >= 7
    assert set(fx[f"act.{tag}.grad_none"].tolist()) == {n_ for n_, p in pol.named_parameters() if p.grad is None}
    np.testing.assert_allclose(pol.bn.running_mean.cpu().numpy(), fx[f"act.{tag}.bn_running_mean"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pol.bn.running_var.cpu().numpy(), fx[f"act.{tag}.bn_running_var"], rtol=RTOL, atol=ATOL)


def _check_dp(tag, cin, pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_dp_policy
    from tests.golden.make_golden import DP_SMALL, PRESAMPLE_SEED
    from tests.util import seeded_fill

    fx = _load("presample_ref.npz")
    pol = build_dp_policy(pcd_npoints=32, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", pre_sample=True,
                          in_channels=cin, **dict(DP_SMALL, pcd_num_classes=20))
    assert seeded_fill(pol, PRESAMPLE_SEED + 1) == float(fx[f"dp.{tag}.wsum"])
    enc = pol.obs_encoder
    # pcd_obs_encoder.py:91-93,103-112: SA layer at the raw width, projector's first convolution at the model's width
    assert tuple(enc.linear.weight.shape) == (cin, 3 + cin) and enc.projector[0].in_channels == 20 and enc.projector[0].out_channels == 24
    pol = pol.to(device).train()
    pre = f"dp.{tag}."
    pcds = {k[len(pre) + 8:]: torch.from_numpy(fx[k]).to(device) for k in fx.files if k.startswith(pre + "in.pcds.")}
    batch = {"obs": {"pcds": pcds, "qpos": torch.from_numpy(fx[pre + "in.qpos"]).to(device)},
             "action": torch.from_numpy(fx[pre + "in.action"]).to(device), "noise": torch.from_numpy(fx[pre + "noise"]).to(device),
             "timesteps": torch.from_numpy(fx[pre + "timesteps"]).to(device)}
    out = pol(batch)
    out["loss"].backward()
    np.testing.assert_allclose(out["loss"].detach().float().cpu().numpy(), fx[pre + "out.loss"], rtol=RTOL, atol=ATOL)
    grads = dict(pol.named_parameters())
    n = 0
    for k in fx.files:
        if k.startswith(pre + "grad."):
            ref = fx[k]
            g = grads[k[len(pre) + 5:]].grad.detach().cpu().numpy()
            assert np.abs(g - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12) + 1e-6, k  # (a bias in front of a BatchNorm has an exactly-zero gradient: both sides hold ~1e-7 of rounding noise)
            n += 1
    assert n >= 12


@pytest.mark.parametrize("tag,cin,use_mask", ACT_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_act_presample_matches_reference_cpu(sa_impl, tag, cin, use_mask):
    from oracle import pointops_cpu

    _check_act(tag, cin, use_mask, pointops_cpu, sa_impl)


@pytest.mark.parametrize("tag,cin", DP_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_dp_presample_matches_reference_cpu(sa_impl, tag, cin):
    from oracle import pointops_cpu

    _check_dp(tag, cin, pointops_cpu, sa_impl)


def test_presample_constructor_contract():
    """Widths follow the reference's constructors; a backbone whose output is not the token width is refused by the builder
    (the reference would fail later, inside the transformer); per-key point-cloud models (pcd_obs_encoder.py:52-63)."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import build_act_policy
    from pointcloudmatters_amd.policy import PointNet
    from pointcloudmatters_amd.policy.diffusion import PCDObsEncoder
    from tests.golden.make_golden import SMALL

    with pytest.raises(ValueError):
        build_act_policy(pcd_npoints=32, pointops=pointops_cpu, pre_sample=True, **SMALL)  # PointNet -> 512 != hidden 48
    meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "wrist": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}},
            "action": {"shape": [7]}}
    enc = PCDObsEncoder(meta, PointNet(6, 24), share_pcd_model=False, pcd_npoints=16, pcd_hidden_dim=24, projector_layers=1,
                        projector_channels=[24, 40, 40], pointops=pointops_cpu)
    assert sorted(enc.key_model_map.keys()) == ["pcds", "wrist"] and enc.key_model_map["pcds"] is not enc.key_model_map["wrist"]
    assert enc.output_shape() == (40 + 9,)
    from pointcloudmatters_amd.bc import make_dp_batch

    a, b = make_dp_batch(2, 60, seed=1)["obs"], make_dp_batch(2, 60, seed=2)["obs"]
    feat = enc({"pcds": a["pcds"], "wrist": b["pcds"], "qpos": a["qpos"][:, :2].reshape(-1, 9)})
    assert feat.shape == (4, 40 + 40 + 9)
    with pytest.raises(RuntimeError):
        PCDObsEncoder({"obs": {"rgb": {"shape": [3, 8, 8], "type": "rgb"}}, "action": {"shape": [7]}}, PointNet(6, 24), pointops=pointops_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,cin,use_mask", ACT_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "torch", "fused"])
def test_act_presample_matches_reference_gpu(hip_device, sa_impl, tag, cin, use_mask):
    import pointcloudmatters_amd.pointops as po

    _check_act(tag, cin, use_mask, po, sa_impl, device=hip_device)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,cin", DP_TAGS)
@pytest.mark.parametrize("sa_impl", ["reference", "torch", "fused"])
def test_dp_presample_matches_reference_gpu(hip_device, sa_impl, tag, cin):
    import pointcloudmatters_amd.pointops as po

    _check_dp(tag, cin, po, sa_impl, device=hip_device)


@pytest.mark.gpu
@pytest.mark.parametrize("cin", [6, 3])
@pytest.mark.parametrize("policy", ["act", "dp"])
def test_presample_yaml_shapes_build_and_train(hip_device, policy, cin):
    """The six PointNet `*_presample*` experiment files at their own widths (ACT: hidden 512 = PointNet's output, 2048 tokens
    from ragged ~4096-point clouds; Diffusion Policy: PointNet head 96, projector [96, 128, 128]) take optimizer steps in the
    mode the benchmark uses for ragged batches, and the fused tokenizer agrees with the reference op order."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, build_dp_policy, clone_batch, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM

    keys = ("color", "coord") if cin == 6 else ("coord",)

    def build(sa_impl):
        torch.manual_seed(3)
        if policy == "act":
            return build_act_policy(pcd_npoints=2048, sa_impl=sa_impl, pre_sample=True, in_channels=cin, dropout=0.0, num_encoder_layers=1,
                                    num_decoder_layers=1).to(hip_device)
        return build_dp_policy(pcd_npoints=2048, sa_impl=sa_impl, pre_sample=True, in_channels=cin, down_dims=(64, 128)).to(hip_device)

    def batch(i):
        if policy == "act":
            b = make_act_batch(2, 4096, seed=40 + i, ragged=True, device=hip_device, feat_keys=keys)
            b["vae_eps"] = torch.randn(2, 32, generator=torch.Generator().manual_seed(i)).to(hip_device)
            return b
        b = make_dp_batch(2, 4096, seed=40 + i, ragged=True, device=hip_device, feat_keys=keys)
        g = torch.Generator().manual_seed(i)
        b["noise"], b["timesteps"] = torch.randn(2, 16, 7, generator=g).to(hip_device), torch.randint(0, 100, (2,), generator=g).to(hip_device)
        return b

    def run(sa_impl, mode):
        pol = build(sa_impl)
        optim = dict(accumulate_grad_batches=1) if policy == "act" else dict(DP_OPTIM)  # (trajectory tolerances below were set on hardware with these)
        tr = BCTrainer(pol, total_steps=50, precision="fp32", device=hip_device, mode=mode, optim=optim)
        return [float(tr.training_step(clone_batch(batch(i)))["loss"]) for i in range(3)]

    ref = run("reference", "eager")
    got = run("fused", "hybrid")
    assert all(np.isfinite(got))
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-4)   # same parameters: the forward must agree to the north_star tolerance
    np.testing.assert_allclose(got, ref, rtol=2e-3)          # after AdamW updates (see test_policy_gpu._compare_mode_runs)
