"""GPU: the RLBench ACT variant through the HIP pointops (incl. the fused set-abstraction kernels) against the reference
fixture, and the RLB workload of bench.py as a short training run."""
import pytest
import torch

from tests.test_rlbench_cpu import build_small_rlbench, check_rlbench, load_rlbench_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sa_impl", ["reference", "fused"])
def test_rlbench_policy_matches_reference_gpu(hip_device, sa_impl):
    import pointcloudmatters_amd.pointops as po

    fx, batch, weights = load_rlbench_fixture(device=hip_device)
    pol = build_small_rlbench(po, sa_impl, weights, device=hip_device)
    run = {k: (dict(v) if isinstance(v, dict) else v) for k, v in batch.items()}
    out = pol(run)
    out["loss"].backward()
    check_rlbench(fx, pol, out, batch)


def test_rlbench_training_step_hybrid_bf16(hip_device):
    from pointcloudmatters_amd.bc import RLBENCH_ACT_OPTIM, BCTrainer, build_rlbench_act_policy, clone_batch, make_act_batch

    torch.manual_seed(0)
    pol = build_rlbench_act_policy(pcd_npoints=128, sa_impl="fused", num_encoder_layers=1, num_decoder_layers=2).to(hip_device)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, mode="hybrid", optim=dict(RLBENCH_ACT_OPTIM, lr=2e-4, accumulate_grad_batches=1))  # (the experiment file accumulates 4: 16 micro-batches would be 4 steps)
    batches = [make_act_batch(4, 400, seed=3 + i, ragged=True, device=hip_device, action_dim=11, qpos_dim=11, goal_cond_dim=512)
               for i in range(2)]
    for b in batches:
        b["actions"][..., -2:] = (b["actions"][..., -2:] > 0).float()
    first = None
    for i in range(16):
        tr.training_step(clone_batch(batches[i % 2]))
        if i == 1:
            first = tr.metrics()["train/loss"]
    last = tr.metrics()["train/loss"]
    assert tr._graph is not None and last == last and last < first, (first, last)


@pytest.mark.parametrize("sa_impl", ["reference", "fused"])
def test_dp_rlbench_policy_matches_reference_modules_gpu(hip_device, sa_impl):
    """The goal-conditioned Diffusion Policy of the RLBench experiments through the HIP kernels against the fixture generated
    from the reference's own modules (loss and seven gradients within 1e-4)."""
    import pointcloudmatters_amd.pointops as po
    from tests.test_rlbench_cpu import build_small_dp_rlbench, check_dp_rlbench, load_dp_rlbench_fixture

    fx, batch, weights = load_dp_rlbench_fixture(device=hip_device)
    pol = build_small_dp_rlbench(po, sa_impl, weights, device=hip_device)
    out = pol(batch)
    out["loss"].backward()
    check_dp_rlbench(fx, pol, out)


def test_dp_rlbench_workload_trains_with_two_micro_batches_per_step(hip_device):
    """bench.py --workload RLBDP in miniature: ragged clouds -> hybrid mode, bf16, 512-d task embedding, accumulate_grad_batches 2
    (configs/exp_rlbench_diffusion_policy/rlbench_model/scratch_pointnet_pcd.yaml:9-13): the optimizer steps every second
    micro-batch and the loss falls on a fixed pair of batches."""
    from pointcloudmatters_amd.bc import RLBENCH_DP_MODEL, RLBENCH_DP_OPTIM, BCTrainer, build_dp_policy, clone_batch, make_dp_batch

    r = RLBENCH_DP_MODEL
    torch.manual_seed(0)
    pol = build_dp_policy(pcd_npoints=256, sa_impl="fused", action_dim=r["action_dim"], qpos_dim=r["qpos_dim"], goal_dim=r["goal_dim"],
                          down_dims=(64, 128, 256)).to(hip_device)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, mode="hybrid", optim=dict(RLBENCH_DP_OPTIM, lr=1e-3))
    assert tr.accumulate == 2
    batches = [make_dp_batch(4, 1024, seed=5 + i, ragged=True, device=hip_device, action_dim=r["action_dim"], qpos_dim=r["qpos_dim"],
                             goal_dim=r["goal_dim"]) for i in range(2)]
    first = None
    for i in range(24):
        tr.training_step(clone_batch(batches[i % 2]))
        if i == 3:
            first = tr.metrics()["train/loss"]
    last = tr.metrics()["train/loss"]
    assert tr.optimizer_steps == 12 and tr.mode == "hybrid" and tr._graph is not None
    assert last == last and last < first, (first, last)
