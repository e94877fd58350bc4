"""GPU, two processes sharing the one device (gloo carries the collectives), for BOTH policies: the data-parallel configuration that
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


def _rank_batches(rank, dev, kind="act"):
    from pointcloudmatters_amd.bc import make_act_batch, make_dp_batch

    if kind == "act_graph":  # graph mode: equal-size clouds, the same layout every step
        return [make_act_batch(2, 300, seed=500 + 10 * i + rank, ragged=False, device=dev, num_queries=10) for i in range(STEPS)]
    if kind == "dp_graph":  # graph mode: equal-size clouds
        return [make_dp_batch(2, 150, seed=700 + 10 * i + rank, ragged=False, device=dev) for i in range(STEPS)]
    if kind == "dp":
        return [make_dp_batch(2, 150, seed=700 + 10 * i + rank, ragged=True, device=dev) for i in range(STEPS)]
    return [make_act_batch(2, 300, seed=500 + 10 * i + rank, ragged=True, device=dev, num_queries=10) for i in range(STEPS)]


def _eps(rank, kind="act"):
    g = torch.Generator().manual_seed(40 + rank)
    if kind.startswith("dp"):  # the DDPM noise of every step (timesteps are fixed per rank below)
        return torch.randn(STEPS, 2, 16, 7, generator=g)
    return torch.randn(STEPS, 2, 8, generator=g)


def _timesteps(rank):
    return torch.tensor([[3 + 40 * rank, 57 + 20 * rank]] * STEPS)


def _concat(b0, b1):
    if "obs" in b0:  # Diffusion-Policy batch: clouds live under obs, sample-major
        obs = _concat({"pcds": b0["obs"]["pcds"], "qpos": b0["obs"]["qpos"]}, {"pcds": b1["obs"]["pcds"], "qpos": b1["obs"]["qpos"]})
        return {"obs": obs, "action": torch.cat([b0["action"], b1["action"]])}
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


def _build(kind):
    from pointcloudmatters_amd.bc import build_act_policy, build_dp_policy
    from pointcloudmatters_amd.bc.configs import DP_OPTIM

    torch.manual_seed(0)
    if kind.startswith("dp"):
        from tests.golden.make_golden import DP_SMALL

        return build_dp_policy(pcd_npoints=32, sa_impl="fused", **DP_SMALL), dict(DP_OPTIM, lr=1e-3)
    return build_act_policy(pcd_npoints=64, sa_impl="fused", **SMALL), dict(accumulate_grad_batches=1, lr=1e-3)


def _train(dev, batches, eps, distributed, kind="act", tsteps=None, mode="hybrid"):
    from pointcloudmatters_amd.bc import BCTrainer, clone_batch

    if kind == "dp_graph" or (kind == "dp" and mode == "flat"):
        # the projector in row layout (opt-in since round 6, PCM_PROJECTOR_ROWS): with it every BatchNorm of the policy is owned by a fused
        # kernel -- what the captured N > 1 chain needs, and what the host-model twin of this test needs (no framework SyncBatchNorm on host tensors)
        from pointcloudmatters_amd.policy import diffusion

        diffusion.PROJECTOR_ROWS = True
    pol, optim = _build(kind)
    tr = BCTrainer(pol.to(dev), total_steps=20, precision="fp32", device=dev, mode=mode, distributed=distributed, optim=optim)
    losses = []
    for i in range(STEPS):
        b = clone_batch(batches[i])
        if kind.startswith("dp"):
            b["noise"], b["timesteps"] = eps[i].to(dev), tsteps[i].to(dev)
        else:
            b["vae_eps"] = eps[i].to(dev)
        losses.append(tr.training_step(b, prefetch=batches[i + 1] if i + 1 < STEPS else None)["loss"].item())
    torch.cuda.synchronize()
    return tr, losses


class _CollectiveLog:
    """Wraps torch.distributed's collectives in the worker: every call is logged with (kind, numel, inside a stream
    capture?), gradient-slab all-reduces are recognised by their storage (views of the flat gradient buffer)."""

    def __init__(self):
        self.calls = []
        self.flat = None
        self._orig = {}

    def install(self):
        for name in ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast", "reduce_scatter_tensor"):
            if hasattr(dist, name):
                self._orig[name] = getattr(dist, name)
                setattr(dist, name, self._wrap(name, self._orig[name]))

    def _wrap(self, name, fn):
        def call(*a, **kw):
            t = a[0] if torch.is_tensor(a[0]) else (a[1] if len(a) > 1 and torch.is_tensor(a[1]) else None)
            is_slab = False
            if t is not None and self.flat is not None and t.is_cuda:
                lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
                is_slab = lo <= t.data_ptr() < hi
            self.calls.append((name, 0 if t is None else t.numel(), bool(torch.cuda.is_current_stream_capturing()), is_slab))
            return fn(*a, **kw)

        return call


def _named(tr):
    """name -> values (the flat buffer's ORDER differs between a staged and an unstaged trainer: compare by name)"""
    return {n: p.detach().float().cpu().numpy() for n, p in tr.policy.named_parameters()}


def _bn_of(tr, kind):
    return tr.policy.obs_encoder.bn if kind.startswith("dp") else tr.policy.bn



def _worker(rank, world, port, q, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        log = _CollectiveLog()
        log.install()
        import pointcloudmatters_amd.bc.trainer as trainer_mod

        real_init = trainer_mod.BCTrainer.__init__

        def init_and_register(self, *a, **kw):  # the flat gradient buffer exists once the trainer is built
            real_init(self, *a, **kw)
            log.flat = self.optimizer.flat_g

        trainer_mod.BCTrainer.__init__ = init_and_register
        mode = "graph" if kind.endswith("_graph") else "hybrid"
        tr, losses = _train(dev, _rank_batches(rank, dev, kind), _eps(rank, kind), distributed=True, kind=kind, tsteps=_timesteps(rank),
                            mode=mode)
        assert tr.mode == mode and tr.distributed and tr.sync_batchnorm and len(tr._stages) == 4
        extra = {}
        if mode == "graph":
            # the step is ONE chain: graphs cut at the synchronised-BatchNorm collectives (one forward, one backward per fused
            # BatchNorm: ACT 6 + 6; Diffusion Policy 8 + 8 since round 5 owns the projector's two) and the non-empty gradient slabs
            from pointcloudmatters_amd.bc import BCTrainer

            assert BCTrainer.all_batchnorms_fused(tr.policy)
            n_bn = sum(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in tr.policy.modules())
            n_slabs = sum(1 for st in tr._stages if st.slab[1] > st.slab[0])
            assert n_bn == (8 if kind.startswith("dp") else 6)
            chain = tr._graph[0]
            assert tr.segmented and len(tr._graph) == 1
            assert chain.n_calls == 2 * n_bn + n_slabs and chain.n_graphs == chain.n_calls + 1, (chain.n_calls, chain.n_graphs)
            extra["chain%d" % rank] = np.asarray([chain.n_graphs, chain.n_calls])
        # what the first multi-GPU execution will do, checked here on one device:
        #  * every collective runs OUTSIDE stream capture (the captured graphs stay collective-free);
        #  * per optimizer step exactly one all-reduce per non-empty gradient slab, each slab exchanged once, in stage order
        assert not any(captured for _, _, captured, _ in log.calls), [c for c in log.calls if c[2]]
        slabs = [c for c in log.calls if c[3] and c[0] == "all_reduce"]
        want = [s.slab[1] - s.slab[0] for s in tr._stages if s.slab[1] > s.slab[0]]
        # (graph mode: the capture pass itself issues one round of slab exchanges -- every rank captures at the same step)
        assert [n for _, n, _, _ in slabs] == want * (STEPS + (mode == "graph")), ([n for _, n, _, _ in slabs], want)
        assert sum(want) == tr.optimizer.flat_g.numel()
        q.put({**extra, "losses%d" % rank: np.asarray(losses), "params%d" % rank: _named(tr),
               "rm%d" % rank: _bn_of(tr, kind).running_mean.detach().cpu().numpy(),
               "ncoll%d" % rank: np.asarray([len(log.calls), len(slabs)])})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["act", "dp", "act_graph", "dp_graph"])
def test_two_ranks_in_hybrid_mode_equal_one_process_on_the_whole_batch(hip_device, kind):
    """kind "act_graph": the same comparison for mode="graph" at N > 1 (round 4) -- the whole step, tokenizer included, replayed
    as a chain of hipGraphs cut at every collective (_graphs.SegmentedCapture); equal-size clouds."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    n_keys = 10 if kind.endswith("_graph") else 8
    for _ in range(3000):
        while not q.empty():
            got.update(q.get())
        if len(got) >= n_keys or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(180)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert len(got) == n_keys
    assert got["ncoll0"].tolist() == got["ncoll1"].tolist()  # both ranks issued the same collectives
    # replicas stay identical
    assert got["params0"].keys() == got["params1"].keys()
    for n in got["params0"]:
        np.testing.assert_array_equal(got["params0"][n], got["params1"][n], err_msg=n)
    np.testing.assert_array_equal(got["rm0"], got["rm1"])
    # ONE process, the concatenated batch (its BatchNorm sees what SyncBN sees across the ranks)
    b0, b1 = _rank_batches(0, hip_device, kind), _rank_batches(1, hip_device, kind)
    whole = [_concat(x, y) for x, y in zip(b0, b1)]
    eps = torch.cat([_eps(0, kind), _eps(1, kind)], dim=1)
    tr, losses = _train(hip_device, whole, eps, distributed=False, kind=kind, tsteps=torch.cat([_timesteps(0), _timesteps(1)], dim=1),
                        mode="graph" if kind.endswith("_graph") else "hybrid")
    mean_losses = (got["losses0"] + got["losses1"]) / 2
    assert mean_losses == pytest.approx(np.asarray(losses), rel=2e-4)
    ref = _named(tr)
    # Adam moves every weight by ~lr per step whatever the gradient's size (an element whose gradient is ~0 may even move
    # the other way after a 1e-7 perturbation), so the yardstick is the UPDATE: per parameter, the two-rank result must sit
    # within a few percent (L2) of the distance the single process travelled from the common initialisation
    init = {n: p.detach().float().numpy() for n, p in _build(kind)[0].named_parameters()}
    index = {id(p): k for k, p in enumerate(tr.optimizer.params)}
    gmax = {n: float(tr.optimizer.g_views[index[id(p)]].abs().max()) if id(p) in index else 0.0 for n, p in tr.policy.named_parameters()}
    gscale = max(gmax.values())
    worst = (0.0, None)
    for n in ref:
        moved = float(np.linalg.norm(ref[n] - init[n]))
        if moved < 1e-6:
            continue  # frozen / unused parameters
        if gmax[n] < 1e-5 * gscale:
            continue  # biases in front of a BatchNorm (the projector's Conv1d biases, PointNet's `final` bias under the SA layer's
            # BatchNorm): their true gradient is exactly zero -- the batch mean is subtracted --, what Adam sees is rounding
            # noise, and it moves them by +-lr in a direction no two runs share
        worst = max(worst, (float(np.linalg.norm(got["params0"][n] - ref[n])) / moved, n))
    assert worst[0] <= 0.05, worst
    # running statistics: equal at the first step; by the third the weights in front of the BatchNorm have drifted by the Adam noise above
    np.testing.assert_allclose(got["rm0"], _bn_of(tr, kind).running_mean.detach().cpu().numpy(), rtol=1e-4, atol=1e-6 if kind == "act" else 1e-5)
