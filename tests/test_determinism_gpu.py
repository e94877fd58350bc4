"""Run-to-run reproducibility of the training step ("deterministic-mode option"): with the sorted
scatter of csrc/sa_scatter.hip (the default, policy/sa_fused.SCATTER_MODE) no float atomic is left on the hot path, so two
executions of the same steps from the same state give bit-identical gradients -- in every execution mode, for both
policies.  The reference cannot offer this: its grouping backward is atomicAdd
(/root/reference/libs/pointops/src/grouping/grouping_cuda_kernel.cu:24)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _act_run(mode, precision, device, ragged, trainer_kw=None, dim_feedforward=32, points=400, decoder_layers=2):
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=512, nhead=8, dim_feedforward=dim_feedforward, num_encoder_layers=2, num_decoder_layers=decoder_layers,
                 dropout=0.1, latent_dim=8, num_queries=10)
    batches = [make_act_batch(3, points, seed=50 + i, ragged=ragged, device=device, num_queries=10) for i in range(2)]
    eps = torch.randn(3, 8, generator=torch.Generator().manual_seed(1)).to(device)
    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=128, sa_impl="fused", **small).to(device)
    tr = BCTrainer(pol, total_steps=20, precision=precision, device=device, mode=mode, optim=dict(accumulate_grad_batches=1, lr=1e-4),
                   **(trainer_kw or {}))
    out = []
    for i in range(3):
        b = clone_batch(batches[i % 2])
        b["vae_eps"] = eps
        loss = tr.training_step(b)["loss"]
        out.append((loss.detach().clone(), tr.optimizer.flat_g.detach().clone()))
    assert tr.mode == mode
    return out, tr.optimizer.flat_p.detach().clone()


@pytest.mark.parametrize("mode,precision,ragged", [("flat", "fp32", True), ("hybrid", "bf16", True), ("graph", "bf16", False),
                                                   ("flat", "bf16", False)])
def test_act_training_steps_are_bit_reproducible(mode, precision, ragged, hip_device):
    """Dropout on (seeded counter hash), three optimizer steps: losses, every flat gradient and the final parameters of two
    independent runs are torch.equal."""
    a, pa = _act_run(mode, precision, hip_device, ragged)
    b, pb = _act_run(mode, precision, hip_device, ragged)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(la, lb), (i, la.item(), lb.item())
        assert torch.equal(ga, gb), (i, (ga - gb).abs().max().item())
    assert torch.equal(pa, pb)


@pytest.mark.parametrize("mode,precision,ragged,ff,staged", [("hybrid", "bf16", True, 3200, None), ("graph", "bf16", False, 3200, None),
                                                             ("flat", "bf16", False, 32, None), ("hybrid", "fp32", True, 3200, None),
                                                             ("hybrid", "bf16", True, 32, True), ("graph", "bf16", False, 32, True)])
def test_batched_closing_reductions_change_no_bit(mode, precision, ragged, ff, staged, hip_device, monkeypatch):
    """policy/deferred.py: the second-level reductions of the fused backward kernels (norm / bias gradients, split-K weight
    gradients) launched together at the end of each backward stage instead of one by one.  Same arithmetic in the same
    order -> losses, every flat gradient and the parameters after three steps are torch.equal with the window on and off;
    a result read before its reduction ran (the failure this guards against) would show up as stale values."""
    from pointcloudmatters_amd.policy import deferred

    monkeypatch.setattr(deferred, "BATCH_WGRADS", False)  # the batched weight gradients agree to rounding only (next test)
    n0 = dict(deferred.STATS)
    # staged=True: the backward stages of the data-parallel runs (one flush per stage, gradients handed over per stage)
    a, pa = _act_run(mode, precision, hip_device, ragged, dict(defer_reductions=True, staged=staged), dim_feedforward=ff, points=1200)
    n1 = dict(deferred.STATS)
    b, pb = _act_run(mode, precision, hip_device, ragged, dict(defer_reductions=False, staged=staged), dim_feedforward=ff, points=1200)
    assert dict(deferred.STATS) == n1 and not deferred.active()
    pushed, launches = n1["pushed"] - n0["pushed"], n1["launches"] - n0["launches"]
    # the window was used, and it batches (four backward stages flush four times: fewer reductions per launch)
    assert pushed >= 10 and launches * (2 if staged else 4) <= pushed, (pushed, launches)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(la, lb), (i, la.item(), lb.item())
        assert torch.equal(ga, gb), (i, (ga - gb).abs().max().item())
    assert torch.equal(pa, pb)


@pytest.mark.parametrize("mode,ragged", [("graph", False), ("hybrid", True), ("flat", True)])
def test_operands_emitted_by_the_producer_change_no_bit(mode, ragged, hip_device, monkeypatch):
    """fused_ops.emit_for: the norm / feed-forward kernels write the bf16 inputs of the projection that consumes their output
    (bf16(out + pos), bf16(out)) in their own launch; that node then skips its add + cast kernel.  Same fp32 values, same
    rounding: three training steps are torch.equal with and without (7-layer decoder, 2-layer encoders)."""
    from pointcloudmatters_amd.policy import fused_ops

    monkeypatch.setattr(fused_ops, "EMIT_OPERANDS", True)
    a, pa = _act_run(mode, "bf16", hip_device, ragged, decoder_layers=3)
    monkeypatch.setattr(fused_ops, "EMIT_OPERANDS", False)
    b, pb = _act_run(mode, "bf16", hip_device, ragged, decoder_layers=3)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(la, lb), (i, la.item(), lb.item())
        assert torch.equal(ga, gb), (i, (ga - gb).abs().max().item())
    assert torch.equal(pa, pb)


@pytest.mark.parametrize("mode,precision,staged", [("graph", "bf16", None), ("hybrid", "bf16", None), ("hybrid", "fp32", None),
                                                   ("hybrid", "bf16", True)])
def test_batched_weight_gradients_agree_and_reproduce(mode, precision, staged, hip_device, monkeypatch):
    """policy/deferred.push_wgrad: the seven decoder layers' weight gradients of one shape as ONE batched product per backward
    stage.  Another GEMM kernel sums in another order, so: against the single products the losses of three steps and every
    gradient agree to rounding (bf16 results: one unit in the last place of some elements); two runs WITH batching are
    bit-identical (the batched kernels are deterministic); and the batching did run."""
    from pointcloudmatters_amd.policy import deferred

    monkeypatch.setattr(deferred, "_EXPECT", {})
    monkeypatch.setattr(deferred, "BATCH_WGRADS", True)
    n0 = dict(deferred.STATS)
    kw = dict(staged=staged)
    a, pa = _act_run(mode, precision, hip_device, mode != "graph", kw, decoder_layers=7)
    n1 = dict(deferred.STATS)
    assert n1["wgrad_batches"] > n0["wgrad_batches"] and n1["wgrads"] - n0["wgrads"] >= 5 * (n1["wgrad_batches"] - n0["wgrad_batches"])
    a2, pa2 = _act_run(mode, precision, hip_device, mode != "graph", kw, decoder_layers=7)
    for (la, ga), (lb, gb) in zip(a, a2):
        assert torch.equal(la, lb) and torch.equal(ga, gb)
    assert torch.equal(pa, pa2)
    monkeypatch.setattr(deferred, "BATCH_WGRADS", False)
    n2 = dict(deferred.STATS)
    b, pb = _act_run(mode, precision, hip_device, mode != "graph", kw, decoder_layers=7)
    assert deferred.STATS["wgrad_batches"] == n2["wgrad_batches"] and deferred.STATS["wgrads"] == n2["wgrads"]
    tol = 2e-2 if precision == "bf16" else 1e-4
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        torch.testing.assert_close(la, lb, rtol=2e-3 if precision == "bf16" else 1e-5, atol=0)
        assert (ga - gb).abs().max() <= tol * gb.abs().max(), (i, (ga - gb).abs().max().item(), gb.abs().max().item())


@pytest.mark.parametrize("mode", ["flat", "hybrid"])
def test_diffusion_policy_training_steps_are_bit_reproducible(mode, hip_device):
    from pointcloudmatters_amd.bc import BCTrainer, build_dp_policy, clone_batch, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    from tests.golden.make_golden import DP_SMALL

    batches = [make_dp_batch(3, 150, seed=40 + i, ragged=True, device=hip_device) for i in range(2)]
    noise = torch.randn(3, 16, 7, generator=torch.Generator().manual_seed(3)).to(hip_device)
    tsteps = torch.tensor([3, 57, 99], device=hip_device)

    def run():
        torch.manual_seed(0)
        pol = build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode=mode, optim=dict(DP_OPTIM, lr=1e-4))
        out = []
        for i in range(3):
            b = clone_batch(batches[i % 2])
            b["noise"], b["timesteps"] = noise, tsteps
            loss = tr.training_step(b)["loss"]
            out.append((loss.detach().clone(), tr.optimizer.flat_g.detach().clone()))
        return out

    a, b = run(), run()
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(la, lb), (i, la.item(), lb.item())
        assert torch.equal(ga, gb), (i, (ga - gb).abs().max().item())
