"""GPU, two processes sharing the one device (gloo carries the collectives): synchronised BatchNorm inside the fused HIP
kernels (policy/sync_bn.py).  Each rank runs PointNet + the fused set-abstraction layer on HALF of a batch; tokens,
running statistics and every gradient must equal the single-process run on the WHOLE batch (gradients summed over the
ranks = the whole-batch gradient: the loss below is a plain sum)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(dev):
    from pointcloudmatters_amd.policy.pointnet import PointNet
    from tests.test_sa_fused_gpu import Owner

    torch.manual_seed(0)
    net = PointNet(in_channels=6, num_classes=0).to(dev).train()
    own = Owner(512, 96, 16).to(dev).train()
    own.sa_impl = "fused"
    with torch.no_grad():
        own.bn.weight.uniform_(-1.0, 1.0)
    return net, own


def _run_part(net, own, clouds, dev):
    """clouds: list of (coord (N,3), feat (N,6)); returns tokens, sum-loss gradients and running statistics."""
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd.policy.sa_layer import set_abstraction

    coord = torch.cat([c for c, _ in clouds]).to(dev)
    feat = torch.cat([f for _, f in clouds]).to(dev)
    sizes = [c.shape[0] for c, _ in clouds]
    off = torch.tensor(sizes).cumsum(0).to(torch.int32)
    off._pcm_host = off.tolist()
    noff = (torch.arange(len(sizes)) + 1).mul(64).to(torch.int32)
    noff._pcm_host = noff.tolist()
    off, noff = off.to(dev), noff.to(dev)
    off._pcm_host, noff._pcm_host = off.tolist(), noff.tolist()
    x = net({"feat": feat})
    _, tok, _ = set_abstraction(own, po, coord, x, off, noff, impl="fused")
    w = torch.linspace(0.5, 1.5, tok.shape[1], device=dev)
    (tok * w).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in list(net.named_parameters()) + [("sa." + k, v) for k, v in own.named_parameters()]}
    stats = {"pn.rm": net.conv5[1].running_mean.clone(), "pn.rv": net.conv5[1].running_var.clone(),
             "sa.rm": own.bn.running_mean.clone(), "sa.rv": own.bn.running_var.clone()}
    return tok.detach(), grads, stats


def _clouds():
    g = torch.Generator().manual_seed(5)
    out = []
    for n in (300, 420, 260, 380):
        out.append((torch.rand(n, 3, generator=g) * 0.8 - 0.4, torch.randn(n, 6, generator=g)))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointcloudmatters_amd.policy.sync_bn import enable_sync_batchnorm

        dev = torch.device("cuda:0")
        net, own = _build(dev)
        holder = torch.nn.ModuleDict({"net": net, "own": own})
        own.fused_batchnorms = lambda: [own.bn]
        enable_sync_batchnorm(holder)
        assert getattr(own.bn, "_pcm_sync", False) and getattr(net.conv1[1], "_pcm_sync", False)
        clouds = _clouds()
        mine = clouds[:2] if rank == 0 else clouds[2:]  # ragged split: 720 vs 640 points
        tok, grads, stats = _run_part(net, own, mine, dev)
        for k in grads:
            dist.all_reduce(grads[k])
        # numpy payloads: a tensor in a queue is a shared-memory handle that dies with this process
        if rank == 0:
            q.put({"tok": tok.cpu().numpy(), "grads": {k: v.cpu().numpy() for k, v in grads.items()},
                   "stats": {k: v.cpu().numpy() for k, v in stats.items()}})
        else:
            q.put({"tok1": tok.cpu().numpy(), "stats1": {k: v.cpu().numpy() for k, v in stats.items()}})
    finally:
        dist.destroy_process_group()


def test_fused_batchnorm_statistics_are_exchanged(hip_device):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(3000):
        while not q.empty():
            got.update(q.get())
        if len(got) >= 5 or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert "tok" in got and "tok1" in got
    net, own = _build(hip_device)
    tok, grads, stats = _run_part(net, own, _clouds(), hip_device)  # the whole batch on one process
    tok2 = torch.cat([torch.from_numpy(got["tok"]), torch.from_numpy(got["tok1"])])
    assert (tok2 - tok.cpu()).abs().max() <= 1e-4 * tok.abs().max().item() + 1e-5
    for k, v in stats.items():  # identical running statistics on both ranks == whole-batch statistics
        torch.testing.assert_close(torch.from_numpy(got["stats"][k]), v.cpu(), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(torch.from_numpy(got["stats1"][k]), v.cpu(), rtol=1e-4, atol=1e-5)
    for k, v in grads.items():
        ref = v.cpu()
        assert (torch.from_numpy(got["grads"][k]) - ref).norm() <= 2e-3 * ref.norm() + 1e-5, k


def _empty_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointcloudmatters_amd.policy import bn_relu
        from pointcloudmatters_amd.policy.sync_bn import enable_sync_batchnorm

        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        bn = torch.nn.BatchNorm1d(64).to(dev).train()
        holder = torch.nn.ModuleDict({"bn": bn})
        holder.fused_batchnorms = lambda: [bn]
        enable_sync_batchnorm(holder)
        y = torch.randn(500, 64, generator=torch.Generator().manual_seed(3)).to(dev)
        mine = (y if rank == 0 else y[:0]).clone().requires_grad_(True)  # rank 1 holds NO rows
        assert bn_relu.supported(mine, bn)
        z = bn_relu.bn_relu(mine, bn)
        (z * 2.0).sum().backward()
        q.put({"z%d" % rank: z.detach().cpu().numpy(), "g%d" % rank: mine.grad.cpu().numpy(), "rm%d" % rank: bn.running_mean.cpu().numpy(),
               "gw%d" % rank: bn.weight.grad.cpu().numpy()})
    finally:
        dist.destroy_process_group()


def test_a_rank_without_rows_still_joins_the_statistics_exchange(hip_device):
    """Ragged data-parallel batches can leave a rank without a single row in front of a synchronised BatchNorm.  It must take
    part in the all_gather / all_reduce with count 0 (like torch's SyncBatchNorm) instead of failing locally while its peers
    block: the rank WITH rows then gets exactly the single-process result."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(1200):
        while not q.empty():
            got.update(q.get())
        if len(got) >= 8 or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert got["z1"].shape == (0, 64) and got["g1"].shape == (0, 64) and not got["gw1"].any()
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(64).to(hip_device).train()
    y = torch.randn(500, 64, generator=torch.Generator().manual_seed(3)).to(hip_device).requires_grad_(True)
    z = torch.relu(bn(y))
    (z * 2.0).sum().backward()
    torch.testing.assert_close(torch.from_numpy(got["z0"]), z.detach().cpu(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(torch.from_numpy(got["g0"]), y.grad.cpu(), rtol=1e-3, atol=1e-5)
    for r in (0, 1):
        torch.testing.assert_close(torch.from_numpy(got["rm%d" % r]), bn.running_mean.cpu(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(torch.from_numpy(got["gw0"]), bn.weight.grad.cpu(), rtol=1e-3, atol=1e-4)


def test_sync_statistics_kernels_match_the_framework_formula(hip_device):
    """pcm_bn_sync_pack_hip / pcm_bn_sync_combine_hip alone (no process group): three pretend ranks with different row counts
    (one of them empty) against the fp64 framework-op formula of sync_bn.combine_forward."""
    import ctypes

    import torch

    from pointcloudmatters_amd import _lib

    L = _lib.load()
    dev = hip_device
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    C, counts = 96, [1000.0, 0.0, 37.0]
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    eps, mom = 1e-3, 0.01
    packs, means, m2s = [], [], []
    for r, n in enumerate(counts):
        rows = int(n)
        for dt in (torch.float32, torch.bfloat16):
            y = (torch.randn(max(rows, 1), C, device=dev) * 2 + 3 + r).to(dt)
            sel = torch.tensor([min(5, max(rows, 1) - 1)], dtype=torch.int32, device=dev)
            shift = y[int(sel[0])].float()
            yy = y[:rows].float()
            sums = torch.stack([(yy - shift).sum(0), ((yy - shift) ** 2).sum(0)]) if rows else torch.zeros(2, C, device=dev)
            pack = torch.empty(2 * C + 1, device=dev)
            assert L.pcm_bn_sync_pack_hip(C, n, sums.data_ptr(), y.data_ptr(), int(dt == torch.bfloat16), sel.data_ptr(), pack.data_ptr(), st) == 0
            if rows:
                torch.testing.assert_close(pack[:C].double(), yy.double().mean(0), rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(pack[C: 2 * C].double(), ((yy.double() - yy.double().mean(0)) ** 2).sum(0), rtol=2e-3, atol=1e-2)
            else:
                assert not pack[: 2 * C].any()
            assert float(pack[2 * C]) == n
        packs.append(pack)  # the bf16 variant of this rank
        means.append(pack[:C].double()), m2s.append(pack[C: 2 * C].double())
    # row 0 when no index is given, zeros for a negative index
    y = torch.randn(4, C, device=dev)
    sums = torch.stack([(y - y[0]).sum(0), ((y - y[0]) ** 2).sum(0)])
    p0, pneg = torch.empty(2 * C + 1, device=dev), torch.empty(2 * C + 1, device=dev)
    assert L.pcm_bn_sync_pack_hip(C, 4.0, sums.data_ptr(), y.data_ptr(), 0, 0, p0.data_ptr(), st) == 0
    torch.testing.assert_close(p0[:C], y.mean(0), rtol=1e-5, atol=1e-6)
    neg = torch.tensor([-1], dtype=torch.int32, device=dev)
    assert L.pcm_bn_sync_pack_hip(C, 4.0, sums.data_ptr(), y.data_ptr(), 0, neg.data_ptr(), pneg.data_ptr(), st) == 0
    torch.testing.assert_close(pneg[:C], sums[0] / 4.0, rtol=1e-6, atol=1e-7)

    gathered = torch.stack(packs).contiguous()
    rm, rv = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    stat, ratio = torch.empty(4, C, device=dev), torch.empty((), device=dev)
    assert L.pcm_bn_sync_combine_hip(3, C, gathered.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, mom, rm.data_ptr(), rv.data_ptr(),
                                     counts[2], stat.data_ptr(), ratio.data_ptr(), st) == 0
    n_r = torch.tensor(counts, dtype=torch.float64, device=dev)[:, None]
    mean_r, m2_r = torch.stack(means), torch.stack(m2s)
    n = n_r.sum()
    mean = (mean_r * n_r).sum(0) / n
    var = ((m2_r + n_r * (mean_r - mean) ** 2).sum(0) / n).clamp_min(0)
    invstd = torch.rsqrt(var + eps)
    a = gamma.double() * invstd
    want = torch.stack([mean, invstd, a, beta.double() - a * mean])
    torch.testing.assert_close(stat.double(), want, rtol=1e-6, atol=1e-6)
    assert abs(float(ratio) - counts[2] / float(n)) < 1e-7
    torch.testing.assert_close(rm, rm0 * (1 - mom) + mom * mean.float(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(rv, rv0 * (1 - mom) + mom * (var * n / (n - 1)).float(), rtol=1e-6, atol=1e-6)
    assert L.pcm_bn_sync_combine_hip(0, C, gathered.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, mom, 0, 0, 1.0, stat.data_ptr(),
                                     ratio.data_ptr(), st) != 0
