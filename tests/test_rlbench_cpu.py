"""CPU: ACTRLBenchPCD against the fixture generated from the reference class (tests/golden/make_golden.py rlbench):
training forward / backward (sigmoid gripper + collision, weighted position loss) and the rollout branch (6-D rotation ->
quaternion).  The same fixture is checked on the GPU through the HIP pointops in tests/test_rlbench_gpu.py."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_rlbench_fixture(device="cpu"):
    fx = np.load(os.path.join(G, "act_rlbench_small.npz"), allow_pickle=False)
    dev = torch.device(device)
    batch = {"pcds": {}}
    for k in fx.files:
        if k.startswith("in.pcds."):
            batch["pcds"][k[len("in.pcds."):]] = torch.from_numpy(fx[k]).to(dev)
        elif k.startswith("in."):
            batch[k[3:]] = torch.from_numpy(fx[k]).to(dev)
    batch["vae_eps"] = torch.from_numpy(fx["eps"]).to(dev)
    weights = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w.")}
    return fx, batch, weights


def build_small_rlbench(pointops, sa_impl, weights, device="cpu"):
    from pointcloudmatters_amd.bc import build_rlbench_act_policy
    from tests.golden.make_golden import RLB_SMALL

    kw = {k: v for k, v in RLB_SMALL.items() if k not in ("rot_type", "collision", "position_loss_weight")}
    pol = build_rlbench_act_policy(pcd_npoints=32, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", **kw)
    pol.load_state_dict(weights, strict=True)
    return pol.to(device).train()


def check_rlbench(fx, pol, out, batch):
    for k in ("a_hat", "mu", "logvar", "loss", "action_loss", "kl_loss"):
        np.testing.assert_allclose(out[k].detach().float().cpu().numpy(), fx[f"out.{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    grads = dict(pol.named_parameters())
    for k in fx.files:
        if k.startswith("grad."):
            g = grads[k[5:]].grad.detach().cpu().numpy()
            ref = fx[k]
            assert np.abs(g - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12) + 1e-6, k
    pol.eval()
    ev = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items() if k not in ("actions", "is_pad", "vae_eps")}
    with torch.no_grad():
        eo = pol(ev)
    np.testing.assert_allclose(eo["a_hat"].float().cpu().numpy(), fx["eval.a_hat"], rtol=2e-4, atol=2e-5)
    q = eo["a_hat"][..., 3:7]
    assert torch.allclose(q.norm(dim=-1), torch.ones_like(q[..., 0]), atol=1e-4) and (q[..., 0] >= 0).all()


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_rlbench_policy_matches_reference_cpu(sa_impl):
    from oracle import pointops_cpu

    fx, batch, weights = load_rlbench_fixture()
    pol = build_small_rlbench(pointops_cpu, sa_impl, weights)
    run = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
    out = pol(run)
    out["loss"].backward()
    check_rlbench(fx, pol, out, batch)


def test_rotation_helpers_roundtrip():
    from pointcloudmatters_amd.policy.rotations import matrix_to_quaternion, rotation_6d_to_matrix

    g = torch.Generator().manual_seed(0)
    d6 = torch.randn(64, 6, generator=g)
    R = rotation_6d_to_matrix(d6)
    eye = torch.eye(3).expand(64, 3, 3)
    assert torch.allclose(R @ R.transpose(-1, -2), eye, atol=1e-5) and torch.allclose(torch.linalg.det(R), torch.ones(64), atol=1e-5)
    q = matrix_to_quaternion(R)
    w, x, y, z = q.unbind(-1)
    R2 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                      2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).view(64, 3, 3)
    assert torch.allclose(R2, R, atol=1e-5)


# ---- RLBench Diffusion Policy: language goal appended to the global condition ------------------------------------------
def load_dp_rlbench_fixture(device="cpu"):
    """dp_rlbench_small.npz (tests/golden/make_golden.py::golden_dp_rlbench): the reference PCDObsEncoder +
    ConditionalUnet1D + LowdimMaskGenerator composed as DiffusionUnetImagePolicy.compute_loss does WITH a task embedding
    (diffusion_unet_image_policy.py:262-266), 11-d action / proprioception."""
    import os

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_rlbench_small.npz"))
    dev = torch.device(device)
    pcds = {k[len("in.pcds."):]: torch.from_numpy(fx[k]).to(dev) for k in fx.files if k.startswith("in.pcds.")}
    batch = {"obs": {"pcds": pcds, "qpos": torch.from_numpy(fx["in.qpos"]).to(dev)}, "action": torch.from_numpy(fx["in.action"]).to(dev),
             "goal": {"task_emb": torch.from_numpy(fx["in.task_emb"]).to(dev)}, "noise": torch.from_numpy(fx["noise"]).to(dev),
             "timesteps": torch.from_numpy(fx["timesteps"]).to(dev)}
    weights = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w.")}
    return fx, batch, weights


def build_small_dp_rlbench(pointops, sa_impl, weights, device="cpu"):
    from pointcloudmatters_amd.bc import RLBENCH_DP_MODEL, build_dp_policy
    from tests.golden.make_golden import DP_SMALL

    r = RLBENCH_DP_MODEL
    pol = build_dp_policy(pcd_npoints=32, pointops=pointops, sa_impl=sa_impl, overlap_sampling=device != "cpu", action_dim=r["action_dim"],
                          qpos_dim=r["qpos_dim"], goal_dim=r["goal_dim"], **DP_SMALL)
    pol.load_state_dict(weights, strict=True)
    return pol.to(device).train()


def check_dp_rlbench(fx, pol, out):
    np.testing.assert_allclose(out["loss"].detach().float().cpu().numpy(), fx["out.loss"], rtol=1e-4, atol=1e-6)
    grads = dict(pol.named_parameters())
    for k in fx.files:
        if k.startswith("grad."):
            g, ref = grads[k[5:]].grad.detach().cpu().numpy(), fx[k]
            assert np.abs(g - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12) + 1e-7, k


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_dp_rlbench_policy_matches_reference_modules_cpu(sa_impl):
    from oracle import pointops_cpu

    fx, batch, weights = load_dp_rlbench_fixture()
    pol = build_small_dp_rlbench(pointops_cpu, sa_impl, weights)
    assert pol.goal_dim == 512
    out = pol(batch)
    out["loss"].backward()
    check_dp_rlbench(fx, pol, out)
    no_goal = {k: v for k, v in batch.items() if k != "goal"}
    with pytest.raises(ValueError):  # a goal-conditioned model refuses a batch without its task embedding
        pol(no_goal)
