"""GPU, two processes sharing the one device (gloo carries the collectives): the data-parallel configuration that
`bench.py --gpus N` runs -- BCTrainer(distributed=True, mode="hybrid"): eager tokenizer with synchronised BatchNorm
inside the fused kernels, captured stages behind the token matrix, gradient slabs exchanged between the replays -- against
ONE process that trains on the concatenated batch.  With SyncBN the two are the same optimisation problem: losses (mean of
the ranks' means) and parameters must agree step after step."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SMALL = dict(hidden_dim=768, nhead=12, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=2, dropout=0.0, latent_dim=8,
             num_queries=10)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_batches(rank, dev):
    from pointcloudmatters_amd.bc import make_act_batch

    return [make_act_batch(2, 300, seed=500 + 10 * i + rank, ragged=True, device=dev, num_queries=10) for i in range(STEPS)]


def _eps(rank):
    return torch.randn(STEPS, 2, 8, generator=torch.Generator().manual_seed(40 + rank))


def _concat(b0, b1):
    out = {}
    for k in b0:
        if k == "pcds":
            n0 = b0["pcds"]["offset"]._pcm_host[-1]
            off = torch.cat([b0["pcds"]["offset"], b1["pcds"]["offset"] + n0])
            off._pcm_host = list(b0["pcds"]["offset"]._pcm_host) + [v + n0 for v in b1["pcds"]["offset"]._pcm_host]
            out["pcds"] = {kk: torch.cat([b0["pcds"][kk], b1["pcds"][kk]]) for kk in ("coord", "grid_coord", "feat")}
            out["pcds"]["offset"] = off
        else:
            out[k] = torch.cat([b0[k], b1[k]])
    return out


def _train(dev, batches, eps, distributed):
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch

    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=64, sa_impl="fused", **SMALL).to(dev)
    tr = BCTrainer(pol, total_steps=20, precision="fp32", device=dev, mode="hybrid", distributed=distributed,
                   optim=dict(accumulate_grad_batches=1, lr=1e-3))
    losses = []
    for i in range(STEPS):
        b = clone_batch(batches[i])
        b["vae_eps"] = eps[i].to(dev)
        losses.append(tr.training_step(b, prefetch=batches[i + 1] if i + 1 < STEPS else None)["loss"].item())
    torch.cuda.synchronize()
    return tr, losses


def _named(tr):
    """name -> values (the flat buffer's ORDER differs between a staged and an unstaged trainer: compare by name)"""
    return {n: p.detach().float().cpu().numpy() for n, p in tr.policy.named_parameters()}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        tr, losses = _train(dev, _rank_batches(rank, dev), _eps(rank), distributed=True)
        assert tr.mode == "hybrid" and tr.distributed and tr.sync_batchnorm and len(tr._stages) == 4
        q.put({"losses%d" % rank: np.asarray(losses), "params%d" % rank: _named(tr),
               "rm%d" % rank: tr.policy.bn.running_mean.detach().cpu().numpy()})
    finally:
        dist.destroy_process_group()


def test_two_ranks_in_hybrid_mode_equal_one_process_on_the_whole_batch(hip_device):
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
        if len(got) >= 6 or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(180)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert len(got) == 6
    # replicas stay identical
    assert got["params0"].keys() == got["params1"].keys()
    for n in got["params0"]:
        np.testing.assert_array_equal(got["params0"][n], got["params1"][n], err_msg=n)
    np.testing.assert_array_equal(got["rm0"], got["rm1"])
    # ONE process, the concatenated batch (its BatchNorm sees what SyncBN sees across the ranks)
    b0, b1 = _rank_batches(0, hip_device), _rank_batches(1, hip_device)
    whole = [_concat(x, y) for x, y in zip(b0, b1)]
    eps = torch.cat([_eps(0), _eps(1)], dim=1)
    tr, losses = _train(hip_device, whole, eps, distributed=False)
    mean_losses = (got["losses0"] + got["losses1"]) / 2
    assert mean_losses == pytest.approx(np.asarray(losses), rel=2e-4)
    ref = _named(tr)
    # Adam moves every weight by ~lr per step whatever the gradient's size (an element whose gradient is ~0 may even move
    # the other way after a 1e-7 perturbation), so the yardstick is the UPDATE: per parameter, the two-rank result must sit
    # within a few percent (L2) of the distance the single process travelled from the common initialisation
    from pointcloudmatters_amd.bc import build_act_policy

    torch.manual_seed(0)
    init = {n: p.detach().float().numpy() for n, p in build_act_policy(pcd_npoints=64, sa_impl="fused", **SMALL).named_parameters()}
    worst = (0.0, None)
    for n in ref:
        moved = float(np.linalg.norm(ref[n] - init[n]))
        if moved < 1e-6:
            continue  # frozen / unused parameters
        worst = max(worst, (float(np.linalg.norm(got["params0"][n] - ref[n])) / moved, n))
    assert worst[0] <= 0.05, worst
    np.testing.assert_allclose(got["rm0"], tr.policy.bn.running_mean.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
