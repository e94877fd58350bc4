"""The TIMED configuration -- bf16 autocast for the transformer / U-Net GEMMs, tokenizer in fp32 (policy/precision.py), fused kernels
-- against the fp32 REFERENCE, tensor by tensor, at 10 %.

Fixture tests/golden/wide_bf16_ref.npz (tests/golden/make_golden.py ``wide_bf16``): the reference's ACTPCD / PCDObsEncoder +
ConditionalUnet1D at the shipped widths on EIGHT samples whose batch seed was searched so that no ReLU gate / arg-max sits within a
bf16 rounding of its kink; stored: inputs, fp32 outputs, a digest of every fp32 gradient, and per tensor the error of the reference's own
bf16 evaluation ("yard.": tokenizer in fp32; "yard_all.": everything under autocast -- up to 59 % off, the reason for the recipe).

Bound per gradient tensor: min(10 %, max(3 x the reference's own bf16 error on that tensor, 5 %)); nothing is exempt (round-4 VERDICT:
"no gradient tensor allowed > 10 % off; U-Net included").  CPU: our classes under torch.autocast("cpu") follow the recipe like the
reference does.  GPU: the fused bf16 path bench.py times."""
import contextlib

import numpy as np
import pytest
import torch

from tests.test_golden_cpu import _load
from tests.util import digest_rel_error, seeded_fill

CAP, FLOOR, YARDS = 0.10, 0.05, 3.0
OUT_RTOL = 2e-2
B = 8


def _digests(fx, prefix):
    out = {}
    for k in fx.files:
        if k.startswith(prefix):
            name, part = k[len(prefix):].rsplit("/", 1)
            out.setdefault(name, {})[part] = fx[k]
    return out


def _act_case(pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_act_policy
    from tests.golden.make_golden import WIDE, WIDE_SEED

    fx = _load("wide_bf16_ref.npz")
    pol = build_act_policy(pcd_npoints=128, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", **WIDE)
    assert seeded_fill(pol, WIDE_SEED) == float(fx["act.wsum"])
    pol = pol.to(device).train()
    batch = {"pcds": {}}
    for k in fx.files:
        if k.startswith("act.in.pcds."):
            batch["pcds"][k[len("act.in.pcds."):]] = torch.from_numpy(fx[k]).to(device)
        elif k.startswith("act.in."):
            batch[k[len("act.in."):]] = torch.from_numpy(fx[k]).to(device)
    batch["vae_eps"] = torch.from_numpy(fx["act.eps"]).to(device)
    return fx, pol, batch


def _dp_case(pointops, sa_impl, device="cpu"):
    from pointcloudmatters_amd.bc import build_dp_policy
    from tests.golden.make_golden import WIDE_DP, WIDE_SEED

    fx = _load("wide_bf16_ref.npz")
    pol = build_dp_policy(pcd_npoints=64, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", **WIDE_DP)
    assert seeded_fill(pol, WIDE_SEED + 1) == float(fx["dp.wsum"])
    pol = pol.to(device).train()
    pcds = {k[len("dp.in.pcds."):]: torch.from_numpy(fx[k]).to(device) for k in fx.files if k.startswith("dp.in.pcds.")}
    batch = {"obs": {"pcds": pcds, "qpos": torch.from_numpy(fx["dp.in.qpos"]).to(device)},
             "action": torch.from_numpy(fx["dp.in.action"]).to(device), "noise": torch.from_numpy(fx["dp.noise"]).to(device),
             "timesteps": torch.from_numpy(fx["dp.timesteps"]).to(device)}
    return fx, pol, batch


def _errors(fx, prefix, pol):
    grads = dict(pol.named_parameters())
    return {name: digest_rel_error(name, grads[name].grad.detach().float().cpu().numpy(), ref) for name, ref in _digests(fx, prefix).items()}


def _judge(fx, which, errs, min_tensors, min_judged=None):
    assert len(errs) >= min_tensors
    bad, judged = [], 0
    for name, (e, scale) in sorted(errs.items()):
        if scale < 1e-6:  # numerically-zero gradients (the decoder's first self-attention acts on an all-zero target)
            continue
        judged += 1
        bound = min(CAP, max(YARDS * float(fx[f"{which}.yard.{name}"]), FLOOR))
        if not e <= bound:
            bad.append((name, round(e, 4), round(bound, 4)))
    assert not bad, bad
    # ACT: the dead second decoder layer (act.py:270) and the first self-attention on an all-zero target have exact-zero gradients
    assert judged >= (min_judged if min_judged is not None else min_tensors - 4), judged
    vals = [e for e, scale in errs.values() if scale >= 1e-6]
    return float(np.median(vals)), float(max(vals))


def test_fixture_documents_why_the_tokenizer_stays_in_fp32():
    """The reference's OWN bf16 evaluation: every tensor within 4 % with the tokenizer in fp32; tens of per cent without."""
    fx = _load("wide_bf16_ref.npz")
    for which in ("act", "dp"):
        yard = [float(fx[k]) for k in fx.files if k.startswith(which + ".yard.")]
        yard_all = [float(fx[k]) for k in fx.files if k.startswith(which + ".yard_all.")]
        assert len(yard) == len(yard_all) >= 90
        assert max(yard) < 0.04 and np.median(yard) < 0.015
        assert max(yard_all) > 0.25  # act 0.58, dp 0.59 when generated


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_act_bf16_recipe_cpu(sa_impl):
    from oracle import pointops_cpu

    fx, pol, batch = _act_case(pointops_cpu, sa_impl)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = pol(batch)
    out["loss"].backward()
    for k in ("a_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src"):
        ref = fx[f"act.out.{k}"]
        assert np.abs(out[k].detach().float().numpy() - ref).max() <= OUT_RTOL * np.abs(ref).max(), k
    assert out["src"].dtype == torch.float32  # the tokens leave the fp32 island as fp32
    med, worst = _judge(fx, "act", _errors(fx, "act.grad.", pol), 90, 75)
    assert med < 0.015, (med, worst)


def test_act_bf16_without_the_recipe_is_far_off_cpu():
    """The switch matters: tokenizer_fp32 = False (autocast everywhere, the behaviour up to round 4) breaks the same bounds."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.policy.precision import set_tokenizer_fp32

    fx, pol, batch = _act_case(pointops_cpu, "reference")
    assert set_tokenizer_fp32(pol, False) == 1
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = pol(batch)
    out["loss"].backward()
    errs = _errors(fx, "act.grad.", pol)
    assert max(e for e, scale in errs.values() if scale >= 1e-6) > 0.25
    with pytest.raises(AssertionError):
        _judge(fx, "act", errs, 90, 75)


def test_dp_bf16_recipe_cpu():
    from oracle import pointops_cpu

    fx, pol, batch = _dp_case(pointops_cpu, "reference")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = pol(batch)
    out["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(fx["dp.out.loss"])) <= OUT_RTOL * float(fx["dp.out.loss"])
    med, worst = _judge(fx, "dp", _errors(fx, "dp.grad.", pol), 100)
    assert med < 0.02, (med, worst)


def test_bf16_mirror_skips_the_fp32_tokenizer():
    """bc/trainer.bf16_consumed_parameters (the weights that get a bf16 mirror / shadow in flat and graph modes, and in the rollout
    wrapper): none of the tokenizer's while it computes in fp32, all of its Linear / Conv1d weights when the switch is off."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc.trainer import bf16_consumed_parameters
    from pointcloudmatters_amd.policy.precision import fp32_tokenizer_parameter_ids, set_tokenizer_fp32

    for case, prefix in ((_act_case, ("backbone.", "linear.", "bn.")), (_dp_case, ("obs_encoder.",))):
        _, pol, _ = case(pointops_cpu, "reference")
        names = {id(p): n for n, p in pol.named_parameters()}
        tok = {names[i] for i in fp32_tokenizer_parameter_ids(pol)}
        assert tok and all(n.startswith(prefix) for n in tok), sorted(tok)[:5]
        mirrored = {names[i] for i in bf16_consumed_parameters(pol)}
        assert mirrored and not (mirrored & tok)
        set_tokenizer_fp32(pol, False)
        assert not fp32_tokenizer_parameter_ids(pol)
        assert {names[i] for i in bf16_consumed_parameters(pol)} > mirrored


# ----------------------------------------------------------------------------------------------------------------- GPU
def _fused_bf16(pol, batch, dev):
    from pointcloudmatters_amd.policy import fused_ops

    ctx = fused_ops.FusedContext(dev)
    with fused_ops.activate(ctx), torch.autocast("cuda", dtype=torch.bfloat16):
        out = pol(batch)
    out["loss"].backward()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mfma_chain", [False, True])
def test_act_bf16_fused_within_ten_percent_of_the_reference_gpu(hip_device, monkeypatch, mfma_chain):
    """mfma_chain: the projection chains through csrc/proj_ln.hip (output projection + residual + norm, in-projections with the
    position add fused in: opt-in, PCM_PROJ_MFMA / PCM_LINEAR_MFMA) -- the same bound holds for them."""
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd import _lib
    from pointcloudmatters_amd.policy import fused_ops

    monkeypatch.setattr(fused_ops, "PROJ_MFMA", mfma_chain)
    monkeypatch.setattr(fused_ops, "LINEAR_MFMA", mfma_chain)
    seen, orig = [], _lib.check

    def check(rc, what, *a, **kw):
        seen.append(what)
        return orig(rc, what, *a, **kw)

    monkeypatch.setattr(_lib, "check", check)
    fx, pol, batch = _act_case(po, "fused", device=hip_device)
    out = _fused_bf16(pol, batch, hip_device)
    # decoder (800 rows) + CVAE encoder (816 rows) sites take the new kernels; the 1048-row encoder site stays with the library
    assert (seen.count("pcm_proj_drln_mfma_forward_hip") >= 4 and seen.count("pcm_linear_mfma_forward_hip") >= 4) == mfma_chain, \
        (seen.count("pcm_proj_drln_mfma_forward_hip"), seen.count("pcm_linear_mfma_forward_hip"))
    for k in ("a_hat", "mu", "logvar", "loss", "action_loss", "kl_loss", "src"):
        ref = fx[f"act.out.{k}"]
        assert np.abs(out[k].detach().float().cpu().numpy() - ref).max() <= OUT_RTOL * np.abs(ref).max(), k
    med, worst = _judge(fx, "act", _errors(fx, "act.grad.", pol), 90, 75)
    print(f"act bf16 fused (mfma_chain={mfma_chain}) vs fp32 reference: median {med:.4f}, worst tensor {worst:.4f}")
    assert med < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["flat", "graph"])
def test_act_bf16_trainer_gradients_within_ten_percent_of_the_reference_gpu(hip_device, mode):
    """The same bound on what BCTrainer(precision="bf16") hands its optimizer (bf16 weight mirrors, deferred reductions, captured
    step): the flat gradient buffer before the first update, unscaled (clip threshold far above the norm)."""
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd.bc import BCTrainer, clone_batch

    fx, pol, batch = _act_case(po, "fused", device=hip_device)
    tr = BCTrainer(pol, total_steps=50, precision="bf16", device=hip_device, mode=mode,
                   optim=dict(accumulate_grad_batches=1, lr=1e-12, weight_decay=0.0, gradient_clip_val=1e9))
    tr.training_step(clone_batch(batch))
    torch.cuda.synchronize()
    opt = tr.optimizer
    index = {id(p): k for k, p in enumerate(opt.params)}
    grads = {n: opt.g_views[index[id(p)]].detach().float().cpu().numpy() for n, p in tr.policy.named_parameters() if id(p) in index}
    errs = {name: digest_rel_error(name, grads[name], ref) for name, ref in _digests(fx, "act.grad.").items()}
    med, worst = _judge(fx, "act", errs, 90, 75)
    print(f"act bf16 trainer[{mode}] vs fp32 reference: median {med:.4f}, worst tensor {worst:.4f}")


@pytest.mark.gpu
def test_dp_bf16_fused_within_ten_percent_of_the_reference_gpu(hip_device):
    import pointcloudmatters_amd.pointops as po

    fx, pol, batch = _dp_case(po, "fused", device=hip_device)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = pol(batch)
    out["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(fx["dp.out.loss"])) <= OUT_RTOL * float(fx["dp.out.loss"])
    med, worst = _judge(fx, "dp", _errors(fx, "dp.grad.", pol), 100)
    print(f"dp bf16 fused vs fp32 reference: median {med:.4f}, worst tensor {worst:.4f}")
    assert med < 0.025
