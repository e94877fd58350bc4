"""GPU: the product policy with the HIP pointops against the golden fixture of the reference's
ACTPCD (fp32, 1e-4 relative), and the training step end to end."""
import pytest
import torch

from tests.test_golden_cpu import build_small_policy, check_against_fixture, load_act_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sa_impl", ["reference", "torch"])
def test_policy_matches_reference_actpcd_gpu(hip_device, sa_impl):
    import pointcloudmatters_amd.pointops as po

    fx, batch, weights = load_act_fixture(device=hip_device)
    pol = build_small_policy(po, sa_impl, weights, device=hip_device)
    out = pol(batch)
    out["loss"].backward()
    check_against_fixture(fx, pol, out)


def test_training_step_runs_and_learns(hip_device):
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=128, sa_impl="torch").to(hip_device)
    tr = BCTrainer(pol, total_steps=200, precision="bf16", device=hip_device, optim=dict(accumulate_grad_batches=1, lr=1e-4))
    batch = make_act_batch(4, 512, seed=1, device=hip_device)
    first = None
    for i in range(12):
        tr.training_step(clone_batch(batch))
        if i == 1:
            first = tr.metrics()["train/loss"]
    last = tr.metrics()["train/loss"]
    assert last == last and last < first


def test_gpu_step_matches_cpu_oracle_step(hip_device):
    """Same weights, same batch, dropout 0: one fp32 training step on the GPU (HIP pointops) and on
    the CPU (oracle pointops) must agree on the loss and on the updated parameters."""
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    torch.manual_seed(3)
    kw = dict(pcd_npoints=64, dropout=0.0, hidden_dim=96, nhead=4, num_encoder_layers=2, num_decoder_layers=2)
    cpu = build_act_policy(pointops=pointops_cpu, sa_impl="reference", **kw)
    gpu = build_act_policy(pointops=po, sa_impl="torch", **kw)
    gpu.load_state_dict(cpu.state_dict())
    gpu.to(hip_device)
    eps = torch.randn(3, 32, generator=torch.Generator().manual_seed(1))
    b_cpu = make_act_batch(3, 300, seed=5, ragged=True)
    b_gpu = make_act_batch(3, 300, seed=5, ragged=True, device=hip_device)
    b_cpu["vae_eps"], b_gpu["vae_eps"] = eps, eps.to(hip_device)
    t_cpu = BCTrainer(cpu, total_steps=100, device="cpu", optim=dict(accumulate_grad_batches=1))
    t_gpu = BCTrainer(gpu, total_steps=100, device=hip_device, optim=dict(accumulate_grad_batches=1))
    l_cpu = t_cpu.training_step(clone_batch(b_cpu))
    l_gpu = t_gpu.training_step(clone_batch(b_gpu))
    torch.testing.assert_close(l_gpu["loss"].cpu(), l_cpu["loss"], rtol=1e-4, atol=1e-5)
    sd_c, sd_g = cpu.state_dict(), gpu.state_dict()
    for k in ("linear.weight", "backbone.conv1.0.weight", "transformer.encoder.layers.0.linear1.weight", "action_head.weight"):
        torch.testing.assert_close(sd_g[k].cpu(), sd_c[k], rtol=1e-3, atol=1e-6)
