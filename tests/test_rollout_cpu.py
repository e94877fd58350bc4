"""Rollout path on the CPU: the host-side pieces and the torch-op form of the policies
against tests/golden/rollout_ref.npz (produced by the reference's own Python, see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from tests.test_golden_cpu import build_small_dp, build_small_policy, load_act_fixture, load_dp_fixture

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_ref.npz")


def load_rollout():
    return np.load(GOLD)


def test_temporal_agg_matches_reference():
    from pointcloudmatters_amd.policy.rollout import TemporalAgg

    fx = load_rollout()
    chunks, want, cut = fx["tagg.chunks"], fx["tagg.out"], int(fx["tagg.reset_after"])
    agg = TemporalAgg(apply=True, action_dim=3, chunk_size=6, k=0.01)
    got = [agg(c) for c in chunks[:cut]]
    agg.reset()
    got += [agg(c) for c in chunks[cut:]]
    assert np.array_equal(np.stack(got), want)  # fp64 host arithmetic: exact
    assert np.array_equal(TemporalAgg(apply=False)(chunks[0]), fx["tagg.noapply"])


def test_ddpm_schedule_matches_oracle():
    from oracle import ddpm_cpu
    from pointcloudmatters_amd.policy.diffusion import DDPMSchedule

    for n_inf in (100, 50, 10):
        s = DDPMSchedule(num_train_timesteps=100)
        s.set_timesteps(n_inf)
        ac = ddpm_cpu.alphas_cumprod(100)
        assert np.array_equal(s.alphas_cumprod.numpy(), ac)
        assert s.timesteps == ddpm_cpu.timesteps(100, n_inf).tolist()
        for t in s.timesteps:
            got = np.array(s.step_coefficients(t), dtype=np.float32)
            want = np.array(ddpm_cpu.step_coefficients(ac, t, n_inf), dtype=np.float32)
            # torch's vectorised CPU sqrt (what diffusers' `x ** 0.5` runs) is not correctly rounded in ~2 % of the
            # entries, numpy's is: allow a few ulp on the derived scalars (the element-wise update itself is tested bit-exactly
            # with shared scalars in test_rollout_gpu.py)
            ulp = np.abs(got.view(np.int32) - want.view(np.int32))
            assert ulp.max() <= 4, (n_inf, t, got, want)
    last = DDPMSchedule(100).step_coefficients(0)
    assert last[2] == 1.0 and last[3] == 0.0 and last[4] == 0.0  # the final update returns the clipped x0


def test_ddpm_step_host_chain_matches_oracle():
    from oracle import ddpm_cpu
    from pointcloudmatters_amd.policy.diffusion import DDPMSchedule

    s = DDPMSchedule(100)
    rng = np.random.default_rng(3)
    eps, xt, nz = (rng.normal(size=(4, 16, 7)).astype(np.float32) for _ in range(3))
    mask = rng.random((4, 16, 7)) < 0.2
    cond = rng.normal(size=(4, 16, 7)).astype(np.float32)
    ac = ddpm_cpu.alphas_cumprod(100)
    for t in (99, 57, 1, 0):
        want = ddpm_cpu.ddpm_step(eps, xt, nz, ddpm_cpu.step_coefficients(ac, t, 100), 1.0, mask, cond)
        got = s.step(torch.from_numpy(eps), t, torch.from_numpy(xt), noise=torch.from_numpy(nz),
                     cond_mask=torch.from_numpy(mask), cond=torch.from_numpy(cond)).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


def with_buffers(weights, fx, prefix):
    w = dict(weights)
    for k in fx.files:
        if k.startswith(prefix):
            w[k[len(prefix):]] = torch.from_numpy(fx[k])
    return w


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_act_rollout_call_matches_reference(sa_impl):
    """ACTPCD in eval mode without "actions": zero latent, BatchNorm running statistics (act.py:177-182)."""
    from oracle import pointops_cpu

    fx = load_rollout()
    _, batch, weights = load_act_fixture()
    pol = build_small_policy(pointops_cpu, sa_impl, with_buffers(weights, fx, "act.buf.")).eval()
    with torch.no_grad():
        out = pol({"qpos": batch["qpos"], "goal_cond": batch["goal_cond"], "pcds": batch["pcds"]})
    assert out["mu"] is None and not out["is_training"]
    np.testing.assert_allclose(out["a_hat"].numpy(), fx["act.a_hat"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["is_pad_hat"].numpy(), fx["act.is_pad_hat"], rtol=1e-4, atol=1e-5)


def test_dp_predict_action_matches_reference_modules():
    """predict_action = reference PCDObsEncoder + ConditionalUnet1D (eval) under the restated DDPM sampler with
    injected noise; 100 chained U-Net calls, so the tolerance is looser than for a single forward."""
    from oracle import pointops_cpu

    fx = load_rollout()
    _, batch, weights = load_dp_fixture()
    pol = build_small_dp(pointops_cpu, "reference", with_buffers(weights, fx, "dp.buf.")).eval()
    noises = [torch.from_numpy(n) for n in fx["dp.noises"]]
    out = pol.predict_action({"obs": {"pcds": batch["obs"]["pcds"], "qpos": batch["obs"]["qpos"]}}, noises=noises)
    assert out["action"].shape == (3, 8, 7) and out["action_pred"].shape == (3, 16, 7)
    np.testing.assert_allclose(out["action_pred"].numpy(), fx["dp.action_pred"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(out["action"].numpy(), fx["dp.action"], rtol=0, atol=2e-3)
    assert np.abs(out["action_pred"].numpy() - fx["dp.action_pred"]).mean() < 1e-4
