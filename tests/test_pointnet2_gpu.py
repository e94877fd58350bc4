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
