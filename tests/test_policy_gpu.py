"""GPU: the product policy with the HIP pointops against the golden fixture of the reference's
ACTPCD (fp32, 1e-4 relative), and the training step end to end."""
import pytest
import torch

from tests.test_golden_cpu import build_small_policy, check_against_fixture, load_act_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sa_impl", ["reference", "torch", "fused"])
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


def _run_mode(mode, hip_device, steps=4, accumulate=1, precision="fp32"):
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    torch.manual_seed(7)
    pol = build_act_policy(pcd_npoints=64, dropout=0.0, hidden_dim=96, nhead=4, num_encoder_layers=2,
                           num_decoder_layers=2, sa_impl="torch").to(hip_device)
    tr = BCTrainer(pol, total_steps=50, precision=precision, device=hip_device, mode=mode,
                   optim=dict(accumulate_grad_batches=accumulate, lr=1e-3))
    batches = [make_act_batch(3, 256, seed=50 + i, device=hip_device) for i in range(2)]
    eps = torch.randn(3, 32, generator=torch.Generator().manual_seed(1)).to(hip_device)
    losses = []
    for i in range(steps * accumulate):
        b = clone_batch(batches[i % 2])
        b["vae_eps"] = eps
        losses.append(tr.training_step(b)["loss"].item())
    return losses, {k: v.detach().clone() for k, v in pol.state_dict().items()}, tr


@pytest.mark.parametrize("accumulate", [1, 2])
def test_flat_and_graph_modes_match_torch_adamw(hip_device, accumulate):
    """FlatAdamW (+ clip, + OneCycle incl. beta1 cycling) and the hipGraph-replayed step must follow
    the same parameter trajectory as torch.optim.AdamW + clip_grad_norm_ + OneCycleLR."""
    l_e, sd_e, _ = _run_mode("eager", hip_device, accumulate=accumulate)
    l_f, sd_f, _ = _run_mode("flat", hip_device, accumulate=accumulate)
    l_g, sd_g, tr = _run_mode("graph", hip_device, accumulate=accumulate)
    assert tr._graph is not None
    for a, b, c in zip(l_e, l_f, l_g):
        assert abs(a - b) <= 2e-4 * abs(a) and abs(a - c) <= 2e-4 * abs(a), (l_e, l_f, l_g)
    for k in sd_e:
        if sd_e[k].dtype.is_floating_point:
            # Adam turns noise-level gradient differences (e.g. the mathematically-zero key-bias
            # gradient of softmax attention) into +-lr updates, so compare per tensor in L2, and bound
            # every element by the largest possible drift (steps * lr).
            ref = sd_e[k].float()
            for other in (sd_f[k], sd_g[k]):
                d = (other.float() - ref)
                slack = 0.05 * 4 * 1e-3 * ref.numel() ** 0.5  # 5% of the total drift steps*lr per element
                assert d.norm() <= 5e-3 * ref.norm() + slack, (k, float(d.norm()), float(ref.norm()))
                assert d.abs().max() <= 2 * 4 * 1e-3 + 1e-6, k
        else:
            assert torch.equal(sd_g[k], sd_e[k]), k  # num_batches_tracked: warm-up must not count


def test_flat_adamw_kernel_vs_torch(hip_device):
    from pointcloudmatters_amd.bc.flat_optim import FlatAdamW
    from pointcloudmatters_amd.bc.schedule import OneCycle

    torch.manual_seed(0)
    shapes = [(513, 7), (64,), (3, 5, 11), (1,), (1000, 33)]
    ref = [torch.nn.Parameter(torch.randn(s, device=hip_device)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt_ref = torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.05)
    sch_ref = torch.optim.lr_scheduler.OneCycleLR(opt_ref, max_lr=3e-3, total_steps=40, pct_start=0.1, div_factor=100.0,
                                                  final_div_factor=1000.0)
    opt = FlatAdamW(mine, OneCycle(3e-3, 40, 0.1, 100.0, 1000.0), weight_decay=0.05, max_norm=0.5)
    for it in range(12):
        grads = [torch.randn(s, device=hip_device) * (3.0 if it % 3 == 0 else 0.01) for s in shapes]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        opt.zero_grad()
        for p, g in zip(mine, grads):
            p.grad.add_(g)
        norm = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt_ref.step()
        sch_ref.step()
        opt.step()
        torch.testing.assert_close(opt.grad_norm[0], norm, rtol=1e-5, atol=1e-7)
        for a, b in zip(mine, ref):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("sa_impl", ["reference", "torch", "fused"])
def test_dp_policy_matches_reference_modules_gpu(hip_device, sa_impl):
    import pointcloudmatters_amd.pointops as po
    from tests.test_golden_cpu import build_small_dp, check_dp_against_fixture, load_dp_fixture

    fx, batch, weights = load_dp_fixture(device=hip_device)
    pol = build_small_dp(po, sa_impl, weights, device=hip_device)
    out = pol(batch)
    out["loss"].backward()
    check_dp_against_fixture(fx, pol, out)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_dp_training_step(hip_device, mode):
    from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, build_dp_policy, clone_batch, make_dp_batch

    torch.manual_seed(0)
    pol = build_dp_policy(pcd_npoints=64, sa_impl="torch", down_dims=(64, 128, 256)).to(hip_device)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, optim=dict(DP_OPTIM, lr=1e-3), mode=mode)
    batch = make_dp_batch(8, 256, seed=2, device=hip_device)
    first = None
    for i in range(25):
        tr.training_step(clone_batch(batch))
        if i == 2:
            first = tr.metrics()["train/loss"]
    last = tr.metrics()["train/loss"]
    assert last == last and last < first, (first, last)


@pytest.mark.parametrize("mode", ["graph", "hybrid"])
def test_graph_mode_coexists_with_rccl_process_group(mode, hip_device):
    """Single-rank RCCL group on the one GPU of this box: the communicator (and its watchdog thread) exists
    while the step is captured into hipGraphs (one per backward stage), and the asynchronous slab all-reduces run
    between the replays -- the exact call sequence bench.py uses for N > 1 (the multi-GPU run itself belongs to the
    driver)."""
    import os

    import torch.distributed as dist

    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        t = torch.ones(4, device=hip_device)
        dist.all_reduce(t)  # forces communicator creation before the capture
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", hidden_dim=768, nhead=4, num_encoder_layers=1,
                               num_decoder_layers=2).to(hip_device)
        tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, mode=mode, staged=True,
                       optim=dict(accumulate_grad_batches=1))
        tr.distributed, tr.world = True, 1  # exercise the exchange branch with a world of one
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)

        dist.all_reduce = counting
        try:
            batch = make_act_batch(4, 256, seed=3, device=hip_device)
            for _ in range(6):
                tr.training_step(clone_batch(batch))
        finally:
            dist.all_reduce = real
        assert tr.mode == mode and tr._graph is not None and len(tr._stages) == 4
        assert len(calls) == 6 * 4 and sum(calls[:4]) == tr.optimizer.flat_g.numel()  # four slabs per step cover the buffer
        m = tr.metrics()
        assert m["train/loss"] == m["train/loss"]
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("how", ["prune_backward", "skip"])
def test_dead_decoder_layer_options_leave_every_parameter_update_unchanged(how, hip_device):
    """ACT reads only decoder output [0]: pruning the backward of layers 1.. (or not evaluating them) must give the same
    loss and the same parameters after optimizer steps as the reference's graph, in which those layers receive exact
    zero gradients (and are still weight-decayed)."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=3, dropout=0.0, latent_dim=8,
                 num_queries=10)
    batch = make_act_batch(2, 256, seed=5, device=hip_device, num_queries=10)
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
    results = {}
    for mode in ("keep", how):
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", dead_decoder_layers=mode, **small).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode="flat", optim=dict(accumulate_grad_batches=1, lr=1e-3))
        b = clone_batch(batch)
        b["vae_eps"] = eps
        loss = tr.training_step(b)["loss"].item()
        opt = tr.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        grads = {n: opt.g_views[index[id(p)]].detach().clone() for n, p in pol.named_parameters() if id(p) in index}
        results[mode] = (loss, grads, {k: v.detach().clone() for k, v in pol.state_dict().items()})
    la, ga, pa = results["keep"]
    lb, gb, pb = results[how]
    assert la == pytest.approx(lb, rel=1e-6)
    assert ga.keys() == gb.keys()
    for k in ga:  # the gradients the optimizer consumed (Adam would amplify noise-level differences: compare them, not weights)
        ref = ga[k]
        assert (gb[k] - ref).norm().item() <= 1e-5 * ref.norm().item() + 1e-9, k
        if ".decoder.layers.1." in k or ".decoder.layers.2." in k:
            assert torch.count_nonzero(ref) == 0 and torch.count_nonzero(gb[k]) == 0, k  # exact zeros either way
    for k in pa:  # ... and those layers are still weight-decayed identically
        if ".decoder.layers.1." in k or ".decoder.layers.2." in k:
            torch.testing.assert_close(pb[k], pa[k], rtol=0, atol=0, msg=k)


def _compare_mode_runs(run_a, run_b, precision="fp32", accumulate=1):
    """Two execution modes of the same maths, compared where the comparison is well-posed.

    Up to the first optimizer step both runs hold the SAME parameters, so losses and gradients may differ only by the
    modes' own arithmetic (fp32: 1e-5 relative -- the kernels are the same and the fused tokenizer is reproducible,
    policy/sa_fused.SCATTER_MODE).  After an update the comparison is no longer one of modes alone: AdamW normalises by
    sqrt(v), so a last-bit difference in a gradient element that is noise (exact-zero-in-theory entries such as softmax key
    biases) becomes a full +-lr parameter change, which feeds the next step.  Later steps are therefore held to a looser
    bound (2e-3: with lr = 1e-5 on weights of magnitude 1e-2 .. 1e-1 the perturbation is at most a few 1e-4 relative per step)."""
    la, ga = run_a[0], run_a[1]
    lb, gb = run_b[0], run_b[1]
    first = accumulate  # micro-batches before the first optimizer step
    tight_l, loose_l = (1e-6, 1e-3) if precision == "fp32" else (2e-2, 2e-2)
    tight_g, loose_g = (1e-5, 2e-3) if precision == "fp32" else (5e-2, 5e-2)
    assert la[:first] == pytest.approx(lb[:first], rel=tight_l), (la, lb)
    assert la[first:] == pytest.approx(lb[first:], rel=loose_l), (la, lb)
    for i, (a, b) in enumerate(zip(ga, gb)):
        tol = tight_g if i < first else loose_g
        assert (a - b).norm().item() <= tol * a.norm().item() + 1e-8, (i, (a - b).norm().item(), a.norm().item())


@pytest.mark.parametrize("precision,accumulate", [("fp32", 1), ("bf16", 1), ("fp32", 2)])
def test_hybrid_mode_matches_flat_mode_on_ragged_batches(precision, accumulate, hip_device):
    """mode="hybrid": eager tokenizer (cloud sizes change every step) + one captured graph for the rest; same maths as
    mode="flat" -- compared through the gradients the optimizer consumes and the losses."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=2, dropout=0.0, latent_dim=8,
                 num_queries=10)
    batches = [make_act_batch(2, 300, seed=90 + i, ragged=True, device=hip_device, num_queries=10) for i in range(3)]
    assert len({b["pcds"]["coord"].shape[0] for b in batches}) == 3  # really ragged
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
    runs = {}
    for mode in ("flat", "hybrid"):
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **small).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision=precision, device=hip_device, mode=mode,
                       optim=dict(accumulate_grad_batches=accumulate, lr=1e-5))
        losses, grads = [], []
        for i in range(4 * accumulate):
            b = clone_batch(batches[i % 3])
            b["vae_eps"] = eps
            losses.append(tr.training_step(b)["loss"].item())
            grads.append(tr.optimizer.flat_g.detach().clone())
        assert tr.mode == mode
        runs[mode] = (losses, grads, pol.bn.running_mean.detach().clone())
    _compare_mode_runs(runs["flat"], runs["hybrid"], precision, accumulate)
    torch.testing.assert_close(runs["flat"][2], runs["hybrid"][2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mode", ["flat", "hybrid"])
def test_prefetched_sampling_gives_identical_steps(mode, hip_device):
    """FPS + kNN computed one batch ahead (trainer.prefetch_sampling) must not change anything: same losses, and each
    batch's indices are consumed exactly once."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0, latent_dim=8,
                 num_queries=10)
    batches = [make_act_batch(2, 300, seed=70 + i, ragged=True, device=hip_device, num_queries=10) for i in range(3)]
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
    runs = []
    for use_prefetch in (False, True):
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **small).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode=mode, optim=dict(accumulate_grad_batches=1, lr=1e-5))
        losses, grads = [], []
        for i in range(6):
            b = clone_batch(batches[i % 3])
            b["vae_eps"] = eps
            losses.append(tr.training_step(b, prefetch=batches[(i + 1) % 3] if use_prefetch else None)["loss"].item())
            grads.append(tr.optimizer.flat_g.detach().clone())
        runs.append((losses, grads))
        assert len(pol.__dict__.get("_prefetched", {})) == (1 if use_prefetch else 0)  # only the batch after the last step is pending
    # a wrong or stale index set would change the loss by percents; see _compare_mode_runs for the two bounds
    _compare_mode_runs(runs[0], runs[1])


def test_hybrid_mode_matches_flat_mode_for_the_diffusion_policy(hip_device):
    from pointcloudmatters_amd.bc import BCTrainer, build_dp_policy, clone_batch, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    from tests.golden.make_golden import DP_SMALL

    batches = [make_dp_batch(3, 150, seed=40 + i, ragged=True, device=hip_device) for i in range(3)]
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(3, 16, 7, generator=g).to(hip_device)
    tsteps = torch.tensor([3, 57, 99], device=hip_device)
    runs = {}
    for mode in ("flat", "hybrid"):
        torch.manual_seed(0)
        pol = build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode=mode, optim=dict(DP_OPTIM, lr=1e-5))
        losses, grads = [], []
        for i in range(4):
            b = clone_batch(batches[i % 3])
            b["noise"], b["timesteps"] = noise, tsteps
            losses.append(tr.training_step(b)["loss"].item())
            grads.append(tr.optimizer.flat_g.detach().clone())
        assert tr.mode == mode
        runs[mode] = (losses, grads)
    _compare_mode_runs(runs["flat"], runs["hybrid"])


@pytest.mark.parametrize("mode,precision", [("flat", "fp32"), ("graph", "bf16"), ("hybrid", "bf16")])
def test_checkpoint_resume_continues_the_same_trajectory(mode, precision, hip_device):
    """trainer.state_dict() -> a fresh trainer -> load_state_dict(): the resumed run produces the losses of the
    uninterrupted one (weights, Adam moments, schedule position and the bf16 weight mirror all restored)."""
    import copy

    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0, latent_dim=8,
                 num_queries=10)
    batch = make_act_batch(2, 256, seed=11, device=hip_device, num_queries=10)
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)

    def make():
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **small).to(hip_device)
        return BCTrainer(pol, total_steps=30, precision=precision, device=hip_device, mode=mode, optim=dict(accumulate_grad_batches=1, lr=1e-3))

    def run(tr, n):
        out = []
        for _ in range(n):
            b = clone_batch(batch)
            b["vae_eps"] = eps
            out.append(tr.training_step(b)["loss"].item())
        return out

    ref = make()
    full = run(ref, 6)
    first = make()
    head = run(first, 3)
    ckpt = copy.deepcopy(first.state_dict())
    resumed = make()
    run(resumed, 1)  # the new trainer has already stepped / captured before the checkpoint arrives
    resumed.load_state_dict(ckpt)
    tail = run(resumed, 3)
    tol = 1e-5 if precision == "fp32" else 2e-2
    assert head == pytest.approx(full[:3], rel=tol)
    assert tail == pytest.approx(full[3:], rel=tol)


def _flat_grads(opt):
    return torch.cat([opt.g_views[k].detach().float().reshape(-1) for k in range(len(opt.params))])


@pytest.mark.parametrize("mode,accumulate", [("graph", 1), ("graph", 2), ("hybrid", 1), ("flat", 1)])
def test_benchmarked_path_end_to_end_against_fp32_eager(mode, accumulate, hip_device):
    """The path bench.py times -- bf16 autocast, fused SA layer, every fused transformer kernel engaged (hidden 512,
    dim_feedforward 32, 8 heads: drln / ffn / small-attention / packed in-projection / bf16 weight mirror), hipGraph
    replay -- against the SAME weights run the reference's way: fp32, eager, reference-order SA layer, framework ops only.
    Indices come from the HIP FPS / kNN in both runs and are checked against the oracle on this batch.  Loss within
    2e-2, gradient direction cosine > 0.999 (bf16 resolution), and graph mode must really have captured."""
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch
    from pointcloudmatters_amd.pointops.query import knn_query_raw

    kw = dict(pcd_npoints=256, dropout=0.0, hidden_dim=512, nhead=8, dim_feedforward=32, num_encoder_layers=2,
              num_decoder_layers=2)
    batches = [make_act_batch(4, 512, seed=200 + i, device=hip_device) for i in range(accumulate)]
    cpu_batches = [make_act_batch(4, 512, seed=200 + i) for i in range(accumulate)]
    eps = torch.randn(4, 32, generator=torch.Generator().manual_seed(1)).to(hip_device)
    # index parity of this very batch: HIP == oracle, bit for bit
    for bg, bc in zip(batches, cpu_batches):
        noff = torch.tensor([256 * (i + 1) for i in range(4)], dtype=torch.int32)
        want = pointops_cpu.farthest_point_sampling(bc["pcds"]["coord"], bc["pcds"]["offset"], noff)
        got = po.farthest_point_sampling(bg["pcds"]["coord"], bg["pcds"]["offset"], noff.to(hip_device))
        assert torch.equal(got.cpu(), want)
        n_p = bc["pcds"]["coord"][want.long()].contiguous()
        wi, _ = pointops_cpu.knn_query_raw(16, bc["pcds"]["coord"], bc["pcds"]["offset"], n_p, noff)
        gi, _ = knn_query_raw(16, bg["pcds"]["coord"], bg["pcds"]["offset"], n_p.to(hip_device), noff.to(hip_device))
        assert torch.equal(gi.cpu(), wi)

    torch.manual_seed(11)
    ref = build_act_policy(sa_impl="reference", **kw).to(hip_device).train()
    torch.manual_seed(11)
    fast = build_act_policy(sa_impl="fused", **kw).to(hip_device).train()
    fast.load_state_dict(ref.state_dict())

    # reference recipe: fp32, eager autograd, gradients of the accumulation window summed with the 1/accumulate loss scale
    from pointcloudmatters_amd.bc.trainer import freeze_unused_parameters

    freeze_unused_parameters(ref)
    ref_losses = []
    for b in batches:
        bb = clone_batch(b)
        bb["vae_eps"] = eps
        out = ref(bb)
        (out["loss"] / accumulate).backward()
        ref_losses.append(out["loss"].item())
    ref_grads = {n: p.grad.detach().float() for n, p in ref.named_parameters() if p.grad is not None}

    tr = BCTrainer(fast, total_steps=100, precision="bf16", device=hip_device, mode=mode,
                   optim=dict(accumulate_grad_batches=accumulate, lr=1e-7))
    losses = []
    for b in batches:
        bb = clone_batch(b)
        bb["vae_eps"] = eps
        losses.append(tr.training_step(bb)["loss"].item())
    assert tr.mode == mode
    if mode in ("graph", "hybrid"):
        assert tr._graph is not None, "the step was not captured into a hipGraph"
        if accumulate > 1:
            assert tr._graph_acc is not None
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-2 * abs(b), (losses, ref_losses)
    opt = tr.optimizer
    index = {id(p): k for k, p in enumerate(opt.params)}
    dot = na = nb = 0.0
    for n, p in fast.named_parameters():
        if id(p) not in index:
            continue
        g = opt.g_views[index[id(p)]].detach().float()
        r = ref_grads.get(n)
        if r is None:
            assert torch.count_nonzero(g) == 0, n
            continue
        dot += float((g * r).sum())
        na += float((g * g).sum())
        nb += float((r * r).sum())
    cos = dot / (na ** 0.5 * nb ** 0.5)
    assert cos > 0.999, cos
    assert abs(na ** 0.5 / nb ** 0.5 - 1.0) < 2e-2, (na, nb)


def test_two_group_flat_adamw_vs_torch(hip_device):
    """timm-style parameter groups (biases / norm weights undecayed): every group must get ITS OWN hyper-parameter row."""
    from pointcloudmatters_amd.bc.flat_optim import FlatAdamW
    from pointcloudmatters_amd.bc.schedule import OneCycle

    torch.manual_seed(0)
    shapes_nd, shapes_d = [(64,), (7,)], [(33, 17), (5, 3, 2)]
    ref_nd = [torch.nn.Parameter(torch.randn(s, device=hip_device)) for s in shapes_nd]
    ref_d = [torch.nn.Parameter(torch.randn(s, device=hip_device)) for s in shapes_d]
    mine_nd = [torch.nn.Parameter(p.detach().clone()) for p in ref_nd]
    mine_d = [torch.nn.Parameter(p.detach().clone()) for p in ref_d]
    opt_ref = torch.optim.AdamW([{"params": ref_nd, "weight_decay": 0.0}, {"params": ref_d, "weight_decay": 0.3}], lr=3e-2,
                                betas=(0.9, 0.95))
    sch_ref = torch.optim.lr_scheduler.OneCycleLR(opt_ref, max_lr=3e-2, total_steps=40, pct_start=0.15, div_factor=100.0,
                                                  final_div_factor=1000.0)
    opt = FlatAdamW([{"params": mine_nd, "weight_decay": 0.0}, {"params": mine_d, "weight_decay": 0.3}],
                    OneCycle(3e-2, 40, 0.15, 100.0, 1000.0), betas=(0.9, 0.95), weight_decay=0.3, max_norm=0.5)
    for it in range(10):
        grads = [torch.randn_like(p) * 0.01 for p in ref_nd + ref_d]
        for p, g in zip(ref_nd + ref_d, grads):
            p.grad = g.clone()
        opt.zero_grad()
        for p, g in zip(mine_nd + mine_d, grads):
            p.grad.add_(g)
        torch.nn.utils.clip_grad_norm_(ref_nd + ref_d, 0.5)
        opt_ref.step()
        sch_ref.step()
        opt.step()
    for a, b in zip(mine_nd + mine_d, ref_nd + ref_d):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)
    # the undecayed group really is undecayed: a fresh optimizer (zero moments) stepping on zero gradients must leave it
    # untouched while the decayed group shrinks by exactly (1 - lr * wd)
    nd = [torch.nn.Parameter(torch.randn(s, device=hip_device)) for s in shapes_nd]
    dd = [torch.nn.Parameter(torch.randn(s, device=hip_device)) for s in shapes_d]
    before_nd, before_d = [p.detach().clone() for p in nd], [p.detach().clone() for p in dd]
    opt2 = FlatAdamW([{"params": nd, "weight_decay": 0.0}, {"params": dd, "weight_decay": 0.3}],
                     OneCycle(3e-2, 40, 0.15, 100.0, 1000.0), betas=(0.9, 0.95), weight_decay=0.3, max_norm=0.5)
    opt2.zero_grad()
    opt2.step()
    for a, b in zip(nd, before_nd):
        assert torch.equal(a.detach(), b)
    for a, b in zip(dd, before_d):
        torch.testing.assert_close(a.detach(), b * (1.0 - opt2.last_lr * 0.3), rtol=1e-6, atol=1e-7)


def test_graph_bf16_accumulation_adds_the_micro_batches(hip_device):
    """graph mode + bf16 hand-off + accumulate_grad_batches=2 (the ACT default): the second micro-batch must ADD its
    gradients to the flat buffer (its own captured graph), not overwrite the first one's."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0, latent_dim=8,
                 num_queries=10)
    batches = [make_act_batch(2, 256, seed=31 + i, device=hip_device, num_queries=10) for i in range(2)]
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
    grads = {}
    for mode in ("flat", "graph"):
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **small).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="bf16", device=hip_device, mode=mode,
                       optim=dict(accumulate_grad_batches=2, lr=1e-7))
        for b in batches:
            bb = clone_batch(b)
            bb["vae_eps"] = eps
            tr.training_step(bb)
        assert tr.mode == mode
        grads[mode] = tr.optimizer.flat_g.detach().clone()
    assert (grads["graph"] - grads["flat"]).norm().item() <= 2e-2 * grads["flat"].norm().item()


def test_dp_hybrid_bf16_full_batch_stays_finite(hip_device):
    """Workload C3R (Diffusion Policy, B=64 x 2 ragged 1024-pt clouds, hybrid mode, bf16): a round-1 bench line recorded
    a NaN loss for it before BatchNorm statistics were accumulated around a reference row; keep it pinned."""
    from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, WORKLOADS, build_dp_policy, clone_batch, make_dp_batch

    wl = WORKLOADS["C3R"]
    torch.manual_seed(1000)
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused", down_dims=(128, 256, 512)).to(hip_device)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=hip_device, mode="hybrid", optim=dict(DP_OPTIM))
    batches = [make_dp_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=True, device=hip_device) for i in range(4)]
    for i in range(24):
        out = tr.training_step(clone_batch(batches[i % 4]))
        if i % 6 == 5:
            assert torch.isfinite(out["loss"]).item(), i
    m = tr.metrics()
    assert m["train/loss"] == m["train/loss"] and m["train/loss"] < 10.0, m
    for n, p in pol.named_parameters():
        assert torch.isfinite(p).all().item(), n


@pytest.mark.parametrize("mode,precision,accumulate", [("flat", "fp32", 1), ("graph", "bf16", 1), ("graph", "bf16", 2), ("hybrid", "bf16", 1),
                                                       ("graph", "fp32", 2)])
def test_backward_stages_give_the_same_gradients(mode, precision, accumulate, hip_device):
    """staged=True (what data-parallel runs use: one partial backward -- and one hipGraph -- per stage, gradient slabs in
    backward order) against the plain single-stage path on one GPU: same losses, same flat gradient (through the
    parameter order, which differs), and in graph / hybrid mode one captured graph per stage."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    small = dict(hidden_dim=512, nhead=8, dim_feedforward=32, num_encoder_layers=2, num_decoder_layers=2, dropout=0.0, latent_dim=8,
                 num_queries=10)
    ragged = mode == "hybrid"
    batches = [make_act_batch(2, 300, seed=60 + i, ragged=ragged, device=hip_device, num_queries=10) for i in range(3)]
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
    runs = {}
    for staged in (False, True):
        torch.manual_seed(0)
        pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **small).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision=precision, device=hip_device, mode=mode, staged=staged,
                       optim=dict(accumulate_grad_batches=accumulate, lr=1e-6))
        assert len(tr._stages) == (4 if staged else 1)
        losses, grads = [], []
        for i in range(3 * accumulate):
            b = clone_batch(batches[i % 3])
            b["vae_eps"] = eps
            losses.append(tr.training_step(b)["loss"].item())
            opt = tr.optimizer
            index = {id(p): k for k, p in enumerate(opt.params)}
            grads.append(torch.cat([opt.g_views[index[id(p)]].detach().reshape(-1).clone() for n, p in pol.named_parameters() if id(p) in index]))
        assert tr.mode == mode
        if mode == "graph":
            assert len(tr._graph) == len(tr._stages)
        if mode == "hybrid":
            assert len(tr._graph) == (3 if staged else 1)
        runs[staged] = (losses, grads)
    tol = 1e-5 if precision == "fp32" else 2e-2
    assert runs[False][0] == pytest.approx(runs[True][0], rel=tol)
    for ga, gb in zip(runs[False][1], runs[True][1]):
        assert (ga - gb).norm().item() <= (1e-4 if precision == "fp32" else 5e-2) * ga.norm().item() + 1e-8


def test_backward_stages_for_the_diffusion_policy(hip_device):
    from pointcloudmatters_amd.bc import BCTrainer, build_dp_policy, clone_batch, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    from tests.golden.make_golden import DP_SMALL

    batches = [make_dp_batch(3, 150, seed=40 + i, ragged=True, device=hip_device) for i in range(3)]
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(3, 16, 7, generator=g).to(hip_device)
    tsteps = torch.tensor([3, 57, 99], device=hip_device)
    runs = {}
    for mode, staged in (("flat", False), ("flat", True), ("hybrid", True)):
        torch.manual_seed(0)
        pol = build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL).to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode=mode, staged=staged, optim=dict(DP_OPTIM, lr=1e-6))
        assert len(tr._stages) == (4 if staged else 1)
        losses, grads = [], []
        for i in range(4):
            b = clone_batch(batches[i % 3])
            b["noise"], b["timesteps"] = noise, tsteps
            losses.append(tr.training_step(b)["loss"].item())
            opt = tr.optimizer
            index = {id(p): k for k, p in enumerate(opt.params)}
            grads.append(torch.cat([opt.g_views[index[id(p)]].detach().reshape(-1).clone() for n, p in pol.named_parameters() if id(p) in index]))
        runs[(mode, staged)] = (losses, grads)
    ref = runs[("flat", False)]
    for key in (("flat", True), ("hybrid", True)):
        assert ref[0] == pytest.approx(runs[key][0], rel=1e-5)
        for ga, gb in zip(ref[1], runs[key][1]):
            assert (ga - gb).norm().item() <= 1e-4 * ga.norm().item() + 1e-8, key


@pytest.mark.parametrize("kind", ["act", "dp"])
def test_graph_mode_with_sampling_outside_the_graph_matches_sampling_inside(kind, hip_device):
    """graph mode keeps FPS / kNN / the SA index pass out of the captured graph (static index buffers filled before each
    replay, computed one batch ahead when the next batch is handed over): same steps as capturing them, with and without
    prefetch, and every batch's indices are consumed exactly once."""
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, build_dp_policy, clone_batch, make_act_batch, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    from tests.golden.make_golden import DP_SMALL

    if kind == "act":
        small = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0, latent_dim=8,
                     num_queries=10)
        batches = [make_act_batch(2, 256, seed=80 + i, device=hip_device, num_queries=10) for i in range(3)]
        eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(1)).to(hip_device)
        extra = {"vae_eps": eps}
        build = lambda: build_act_policy(pcd_npoints=64, sa_impl="fused", **small)
        optim = dict(accumulate_grad_batches=1, lr=1e-5)
    else:
        batches = [make_dp_batch(3, 128, seed=40 + i, device=hip_device) for i in range(3)]
        g = torch.Generator().manual_seed(3)
        extra = {"noise": torch.randn(3, 16, 7, generator=g).to(hip_device), "timesteps": torch.tensor([3, 57, 99], device=hip_device)}
        build = lambda: build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL)
        optim = dict(DP_OPTIM, lr=1e-5)
    runs = {}
    for external, use_prefetch in ((False, False), (True, False), (True, True)):
        torch.manual_seed(0)
        pol = build().to(hip_device)
        tr = BCTrainer(pol, total_steps=20, precision="fp32", device=hip_device, mode="graph", optim=optim, external_sampling=external)
        losses = []
        for i in range(6):
            b = clone_batch(batches[i % 3])
            b.update(extra)
            losses.append(tr.training_step(b, prefetch=batches[(i + 1) % 3] if use_prefetch else None)["loss"].item())
        assert tr.mode == "graph" and tr._static_sampling == external
        owner = pol if kind == "act" else pol.obs_encoder
        assert len(owner.__dict__.get("_prefetched", {})) == (1 if use_prefetch else 0)  # only the batch after the last step
        runs[(external, use_prefetch)] = losses
    base = runs[(False, False)]
    assert all(l == l for l in base)
    for key in ((True, False), (True, True)):  # same parameters up to the first update: tight; later steps: _compare_mode_runs
        assert runs[key][:1] == pytest.approx(base[:1], rel=1e-6), key
        assert runs[key][1:] == pytest.approx(base[1:], rel=1e-3), key
