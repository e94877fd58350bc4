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
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, mode="hybrid", optim=dict(RLBENCH_ACT_OPTIM, lr=2e-4))
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
