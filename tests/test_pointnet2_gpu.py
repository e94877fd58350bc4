"""Hierarchical SA encoders (policy/pointnet2.py; no reference counterpart): the fused HIP composition against the
SAME composition evaluated on the CPU with the oracle's pointops, forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(device, ragged=True):
    from pointcloudmatters_amd.bc import make_act_batch

    return make_act_batch(3, 700, seed=21, ragged=ragged, device=device)["pcds"]


def _grads(mod):
    return {k: p.grad.detach().cpu().clone() for k, p in mod.named_parameters() if p.grad is not None}


def _close(a, b, tol=2e-4):
    assert (a - b).abs().max().item() <= tol * (b.abs().max().item() + 1e-12) + 1e-6


def test_pointnet2_encoder_and_fp_match_cpu_oracle_composition():
    from oracle import pointops_cpu
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.policy.pointnet2 import FeaturePropagation, PointNet2Encoder

    torch.manual_seed(0)
    stages = ((128, 16, 32), (32, 8, 64))
    cpu = PointNet2Encoder(6, 16, stages, pointops=pointops_cpu, sa_impl="reference")
    cpu_fp = FeaturePropagation(64, 32, 24, pointops=pointops_cpu)
    gpu = PointNet2Encoder(6, 16, stages, pointops=pointops, sa_impl="fused")
    gpu_fp = FeaturePropagation(64, 32, 24, pointops=pointops)
    gpu.load_state_dict(cpu.state_dict()), gpu_fp.load_state_dict(cpu_fp.state_dict())
    gpu, gpu_fp = gpu.to(DEV), gpu_fp.to(DEV)
    outs = []
    for enc, fp, dev in ((cpu, cpu_fp, "cpu"), (gpu, gpu_fp, DEV)):
        levels = enc(_batch(dev), return_all=True)
        dense = fp(levels[1], levels[2])  # propagate the coarsest features back onto stage-1 points
        (levels[2][1].square().mean() + dense.square().mean()).backward()
        outs.append((levels, dense))
    (lc, dc), (lg, dg) = outs
    for (pc, xc, oc), (pg, xg, og) in zip(lc, lg):
        assert torch.equal(pc, pg.cpu()) and oc.tolist() == og.tolist()  # FPS picks are bit-exact
        _close(xg.detach().float().cpu(), xc.detach())
    _close(dg.detach().cpu(), dc.detach())
    gc, gg = _grads(cpu), _grads(gpu)
    assert gc.keys() == gg.keys()
    for k in gc:
        _close(gg[k], gc[k], 5e-4)
    for k, v in _grads(cpu_fp).items():
        _close(_grads(gpu_fp)[k], v, 5e-4)


def test_patch_tokenizer_shapes_and_parity():
    from oracle import pointops_cpu
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.policy.pointnet2 import PatchTokenizer

    torch.manual_seed(1)
    cpu = PatchTokenizer(6, 16, 32, 48, pointops=pointops_cpu, sa_impl="reference")
    gpu = PatchTokenizer(6, 16, 32, 48, pointops=pointops, sa_impl="fused")
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.to(DEV)
    tc, pc = cpu(_batch("cpu"))
    tg, pg = gpu(_batch(DEV))
    assert tg.shape == (3, 16, 48) and pg.shape == (3, 16, 48)
    _close(tg.detach().cpu(), tc.detach())
    _close(pg.cpu(), pc, 1e-5)


@pytest.mark.parametrize("n_points", [2048, 4096])
def test_msg_and_pointnext_match_cpu_oracle_composition_at_config_sizes(n_points):
    """PointNet++ MSG stage and the PointNeXt backbone (InvResMLP blocks) at the cloud sizes of BASELINE configs[3] / [4]:
    fused HIP composition == the same modules on the CPU with the oracle's pointops, forward and backward."""
    from oracle import pointops_cpu
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.bc import make_act_batch
    from pointcloudmatters_amd.policy.pointnet2 import PointNeXtBackbone, SAStageMSG

    torch.manual_seed(2)
    scales = ((16, 0.06, 32), (32, None, 64))
    cpu_msg = SAStageMSG(32, n_points // 4, scales, pointops=pointops_cpu, sa_impl="reference")
    gpu_msg = SAStageMSG(32, n_points // 4, scales, pointops=pointops, sa_impl="fused")
    cpu_nx = PointNeXtBackbone(6, 32, 2, 16, 32, pointops=pointops_cpu, sa_impl="reference")
    gpu_nx = PointNeXtBackbone(6, 32, 2, 16, 32, pointops=pointops, sa_impl="fused")
    gpu_msg.load_state_dict(cpu_msg.state_dict()), gpu_nx.load_state_dict(cpu_nx.state_dict())
    gpu_msg, gpu_nx = gpu_msg.to(DEV), gpu_nx.to(DEV)
    outs = []
    for nx, msg, dev in ((cpu_nx, cpu_msg, "cpu"), (gpu_nx, gpu_msg, DEV)):
        pcd = make_act_batch(2, n_points, seed=33, ragged=True, device=dev)["pcds"]
        x = nx(pcd)
        n_p, tok, n_o = msg(pcd["coord"], x, pcd["offset"])
        tok.square().mean().backward()
        outs.append((x.detach().float().cpu(), n_p.cpu(), tok.detach().float().cpu()))
    (xc, pc, tc), (xg, pg, tg) = outs
    assert torch.equal(pc, pg)  # FPS picks bit-exact
    _close(xg, xc, 5e-4)
    _close(tg, tc, 1e-3)
    for cm, gm in ((cpu_nx, gpu_nx), (cpu_msg, gpu_msg)):
        gc, gg = _grads(cm), _grads(gm)
        assert gc.keys() == gg.keys()
        for k in gc:
            assert (gg[k] - gc[k]).norm() <= 5e-3 * gc[k].norm() + 1e-7, k


def test_patchbert_obs_encoder_matches_cpu_oracle_composition():
    from oracle import pointops_cpu
    from pointcloudmatters_amd import pointops
    from pointcloudmatters_amd.bc import make_act_batch
    from pointcloudmatters_amd.policy.pointnet2 import PatchBertObsEncoder

    meta = {"obs": {"pcds": {"shape": [6], "type": "pcd"}, "qpos": {"shape": [9], "type": "low_dim"}}}
    torch.manual_seed(3)
    kw = dict(num_groups=64, group_size=32, hidden_dim=96, depth=2, nhead=4, out_channels=48)  # hidden_dim // 3 must be even (sine embedding)
    cpu = PatchBertObsEncoder(meta, pointops=pointops_cpu, sa_impl="reference", **kw)
    gpu = PatchBertObsEncoder(meta, pointops=pointops, sa_impl="fused", **kw)
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.to(DEV)
    q = torch.randn(2, 9)
    oc = cpu({"pcds": make_act_batch(2, 4096, seed=34, ragged=True)["pcds"], "qpos": q})
    og = gpu({"pcds": make_act_batch(2, 4096, seed=34, ragged=True, device=DEV)["pcds"], "qpos": q.to(DEV)})
    assert og.shape == (2, 48 + 9)
    _close(og.detach().cpu(), oc.detach(), 1e-3)
    oc.square().mean().backward(), og.square().mean().backward()
    gc, gg = _grads(cpu), _grads(gpu)
    for k in gc:
        assert (gg[k] - gc[k]).norm() <= 5e-3 * gc[k].norm() + 1e-7, k


@pytest.mark.parametrize("workload", ["C4N", "C5B"])
def test_hierarchical_workloads_train(workload):
    """The two workloads that use the hierarchical encoders (PointNeXt + ACT, PointBERT patches + Diffusion Policy) run in
    the benchmarked trainer mode and learn on a fixed batch."""
    from pointcloudmatters_amd.bc import (DP_OPTIM, BCTrainer, WORKLOADS, build_act_policy, build_dp_policy, clone_batch,
                                          make_act_batch, make_dp_batch)

    wl = WORKLOADS[workload]
    torch.manual_seed(0)
    if wl["policy"] == "dp":
        pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused", obs_encoder=wl["obs_encoder"], down_dims=(64, 128, 256)).to(DEV)
        tr = BCTrainer(pol, total_steps=100, precision="bf16", device=DEV, mode="graph", optim=dict(DP_OPTIM, lr=1e-3))
        batch = make_dp_batch(4, 1024, seed=2, device=DEV)
    else:
        pol = build_act_policy(pcd_npoints=256, sa_impl="fused", backbone=wl["backbone"], num_encoder_layers=1, num_decoder_layers=2).to(DEV)
        tr = BCTrainer(pol, total_steps=100, precision="bf16", device=DEV, mode="graph", optim=dict(accumulate_grad_batches=1, lr=2e-4))
        batch = make_act_batch(2, 512, seed=2, device=DEV)
    first = None
    for i in range(20):
        tr.training_step(clone_batch(batch))
        if i == 2:
            first = tr.metrics()["train/loss"]
    last = tr.metrics()["train/loss"]
    assert tr._graph is not None and last == last and last < first, (first, last)
