"""Rollout path on the MI355X: the fused DDPM update (csrc/ddpm.hip) bit-exact against oracle/ddpm_cpu.py, the
policies' inference calls against the reference-generated fixture, and hipGraph replay == eager."""
import numpy as np
import pytest
import torch

from tests.test_golden_cpu import build_small_dp, build_small_policy, load_act_fixture, load_dp_fixture
from tests.test_rollout_cpu import load_rollout, with_buffers

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n", [1, 7, 112, 4096 + 3, 1 << 20])
@pytest.mark.parametrize("eps_bf16", [False, True])
def test_ddpm_step_kernel_bit_exact(n, eps_bf16):
    from oracle import ddpm_cpu
    from pointcloudmatters_amd.policy.diffusion import DDPMSchedule

    s = DDPMSchedule(100)
    rng = np.random.default_rng(n + int(eps_bf16))
    eps = (2 * rng.normal(size=n)).astype(np.float32)
    xt, nz, cond = (rng.normal(size=n).astype(np.float32) for _ in range(3))
    mask = rng.random(n) < 0.3
    eps_t = torch.from_numpy(eps).to(DEV)
    if eps_bf16:
        eps_t = eps_t.bfloat16()
        eps = eps_t.float().cpu().numpy()
    for t in (99, 64, 1, 0):
        coef = tuple(np.float32(c) for c in s.step_coefficients(t))  # shared scalars: the update itself is under test
        for use_mask in (False, True):
            want = ddpm_cpu.ddpm_step(eps, xt, nz, coef, 1.0, mask if use_mask else None, cond)
            got = s.step(eps_t, t, torch.from_numpy(xt).to(DEV), noise=torch.from_numpy(nz).to(DEV),
                         cond_mask=torch.from_numpy(mask).to(DEV) if use_mask else None,
                         cond=torch.from_numpy(cond).to(DEV) if use_mask else None)
            assert np.array_equal(got.cpu().numpy(), want), (t, use_mask)
    # in-place form (prev aliases x_t), and the draw-inside form produces finite values of the right spread
    x = torch.from_numpy(xt).to(DEV)
    keep = x.clone()
    out = s.step(eps_t, 64, x, noise=torch.from_numpy(nz).to(DEV), out=x)
    assert out.data_ptr() == x.data_ptr()
    assert torch.equal(out, s.step(eps_t, 64, keep, noise=torch.from_numpy(nz).to(DEV)))
    assert torch.isfinite(s.step(eps_t, 64, keep)).all()


def test_ddpm_step_kernel_empty_and_errors():
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    assert L.pcm_ddpm_step_hip(0, 0, 0, 0, 0, 0, 0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0, 0) == 0
    x = torch.zeros(4, device=DEV)
    m = torch.zeros(4, dtype=torch.uint8, device=DEV)
    assert L.pcm_ddpm_step_hip(4, 0, x.data_ptr(), x.data_ptr(), 0, m.data_ptr(), 0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0,
                               x.data_ptr(), 0) != 0  # mask without cond


@pytest.mark.parametrize("sa_impl", ["reference", "fused"])
def test_act_rollout_call_matches_reference_gpu(sa_impl):
    from pointcloudmatters_amd import pointops

    fx = load_rollout()
    _, batch, weights = load_act_fixture(DEV)
    pol = build_small_policy(pointops, sa_impl, with_buffers(weights, fx, "act.buf."), DEV).eval()
    with torch.no_grad():
        out = pol({"qpos": batch["qpos"], "goal_cond": batch["goal_cond"], "pcds": dict(batch["pcds"])})
    np.testing.assert_allclose(out["a_hat"].cpu().numpy(), fx["act.a_hat"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out["is_pad_hat"].cpu().numpy(), fx["act.is_pad_hat"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("sa_impl", ["reference", "fused"])
def test_dp_predict_action_matches_reference_gpu(sa_impl):
    from pointcloudmatters_amd import pointops

    fx = load_rollout()
    _, batch, weights = load_dp_fixture(DEV)
    pol = build_small_dp(pointops, sa_impl, with_buffers(weights, fx, "dp.buf."), DEV).eval()
    noises = [torch.from_numpy(n).to(DEV) for n in fx["dp.noises"]]
    out = pol.predict_action({"obs": {"pcds": dict(batch["obs"]["pcds"]), "qpos": batch["obs"]["qpos"]}}, noises=noises)
    got = out["action_pred"].cpu().numpy()
    np.testing.assert_allclose(got, fx["dp.action_pred"], rtol=0, atol=5e-3)  # 100 chained U-Net calls
    assert np.abs(got - fx["dp.action_pred"]).mean() < 2e-4
    np.testing.assert_allclose(out["action"].cpu().numpy(), fx["dp.action"], rtol=0, atol=5e-3)


def test_graphed_act_replay_equals_eager():
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.bc import make_act_batch
    from pointcloudmatters_amd.policy.rollout import graphed_act
    from tests.golden.make_golden import SMALL

    fx = load_rollout()
    _, _, weights = load_act_fixture(DEV)
    pol = build_small_policy(pointops, "fused", with_buffers(weights, fx, "act.buf."), DEV).eval()

    def obs(seed):
        b = make_act_batch(2, 200, seed=seed, device=DEV, num_queries=SMALL["num_queries"])
        return {"qpos": b["qpos"], "goal_cond": b["goal_cond"], "pcds": b["pcds"]}

    first, second = obs(1), obs(2)
    runner = graphed_act(pol, first)
    for o in (first, second, first):
        got = runner(o).float().clone()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want = pol(dict(o, pcds=dict(o["pcds"])))["a_hat"].float()
        torch.testing.assert_close(got, want, rtol=0, atol=0)  # same kernels, same order: identical


def test_graphed_dp_samples_fresh_noise_per_replay():
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.bc import make_dp_batch
    from pointcloudmatters_amd.policy.rollout import graphed_dp

    fx = load_rollout()
    _, _, weights = load_dp_fixture(DEV)
    pol = build_small_dp(pointops, "fused", with_buffers(weights, fx, "dp.buf."), DEV).eval()
    b = make_dp_batch(2, 150, seed=5, device=DEV)
    example = {"obs": {"pcds": b["obs"]["pcds"], "qpos": b["obs"]["qpos"][:, :2].contiguous()}}
    runner = graphed_dp(pol, example)
    a = runner(example)["action_pred"].clone()
    c = runner(example)["action_pred"].clone()
    assert a.shape == (2, 16, 7) and torch.isfinite(a).all() and a.abs().max() <= 1.0
    assert not torch.equal(a, c)  # the Philox offset advances with every replay
    assert runner.static_out["action"].shape == (2, 8, 7)
