"""CPU, world_size 2, gloo: the data-parallel path of the BC trainer.

* mode="eager": torch DistributedDataParallel around the policy (what bench.py --mode eager runs
  over RCCL), pointops = CPU oracle.
* mode="flat":  the product's own data parallelism -- ONE all-reduce of the flat gradient buffer and
  the 1/world scale folded into the optimizer -- with a host stand-in for the HIP FlatAdamW kernels
  (same interface, torch ops) so that the trainer logic runs without a GPU.
Checks: replicas stay bit-identical, and two ranks fed the same batch reproduce the single-process
trajectory (gradient mean == single gradient when BN statistics coincide).
"""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class HostFlatAdamW:
    """Host stand-in with FlatAdamW's interface (flat_g, zero_grad, step) built on torch.optim.AdamW."""

    def __init__(self, params, schedule, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=0.0, grad_scale=1.0):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat_g = torch.zeros(n)
        o = 0
        for p in self.params:
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)
            o += p.numel()
        self.opt = torch.optim.AdamW(self.params, lr=1.0, betas=betas, eps=eps, weight_decay=weight_decay)
        self.schedule, self.max_norm, self.grad_scale, self.step_count = schedule, max_norm, grad_scale, 0

    def zero_grad(self):
        self.flat_g.zero_()

    def step(self):
        lr, mom = self.schedule.at(self.step_count)
        g = self.opt.param_groups[0]
        g["lr"] = lr
        if mom is not None:
            g["betas"] = (mom, g["betas"][1])
        self.flat_g.mul_(self.grad_scale)
        if self.max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
        self.opt.step()
        self.step_count += 1

    def state_dict(self):
        return {}


def _worker(rank, world, port, mode, same_data, out_q, staged=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pointops_cpu
        from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

        torch.manual_seed(0)  # identical initial weights on every rank (DDP would broadcast rank 0's)
        pol = build_act_policy(pcd_npoints=32, pointops=pointops_cpu, sa_impl="reference", dropout=0.0, hidden_dim=48,
                               nhead=4, num_encoder_layers=1, num_decoder_layers=2, num_queries=10)
        tr = BCTrainer(pol, total_steps=50, device="cpu", distributed=True, sync_batchnorm=False, mode=mode,
                       optim=dict(accumulate_grad_batches=2, lr=1e-3),
                       flat_optimizer_cls=HostFlatAdamW if mode == "flat" else None, staged=staged)
        assert tr.distributed and tr.world == world
        if mode == "flat":
            assert len(tr._stages) == (1 if staged is False else 4)  # decoder | encoder | CVAE + projections | tokenizer
            slabs = [s.slab for s in tr._stages]
            assert slabs[0][0] == 0 and slabs[-1][1] == tr.optimizer.flat_g.numel()
            assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))  # contiguous, in backward order
        eps = torch.randn(2, 32, generator=torch.Generator().manual_seed(3))
        for it in range(4):  # 2 optimizer steps, accumulate 2
            seed = 100 + it if same_data else 100 + it * world + rank
            b = make_act_batch(2, 120, seed=seed, num_queries=10)
            b["vae_eps"] = eps
            tr.training_step(clone_batch(b))
        m = tr.metrics()
        flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            out_q.put({"replicas_equal": all(torch.equal(gathered[0], g) for g in gathered), "params": flat.numpy().copy(),
                       "loss": m["train/loss"], "opt_steps": tr.optimizer_steps})
    finally:
        dist.destroy_process_group()


def _run(mode, same_data, staged=None):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, same_data, q, staged)) for r in range(2)]
    for p in procs:
        p.start()
    res = None
    for _ in range(3000):  # drain before join: a large item blocks the writer until it is read
        if not q.empty():
            res = q.get()
            break
        if any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert res is not None
    return res


def _single(mode):
    from oracle import pointops_cpu
    from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch

    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=32, pointops=pointops_cpu, sa_impl="reference", dropout=0.0, hidden_dim=48,
                           nhead=4, num_encoder_layers=1, num_decoder_layers=2, num_queries=10)
    tr = BCTrainer(pol, total_steps=50, device="cpu", mode="eager", optim=dict(accumulate_grad_batches=2, lr=1e-3))
    eps = torch.randn(2, 32, generator=torch.Generator().manual_seed(3))
    for it in range(4):
        b = make_act_batch(2, 120, seed=100 + it, num_queries=10)
        b["vae_eps"] = eps
        tr.training_step(clone_batch(b))
    return torch.cat([p.detach().reshape(-1) for p in pol.parameters()]), tr.metrics()["train/loss"]


@pytest.mark.parametrize("mode", ["eager", "flat"])
def test_two_ranks_sharded_data_stay_in_sync(mode):
    r = _run(mode, same_data=False)
    assert r["replicas_equal"] and r["opt_steps"] == 2 and r["loss"] == r["loss"]


@pytest.mark.parametrize("mode", ["eager", "flat"])
def test_two_ranks_same_data_match_single_process(mode):
    r = _run(mode, same_data=True)
    want, loss = _single(mode)
    assert r["replicas_equal"]
    assert abs(r["loss"] - loss) <= 1e-5 * abs(loss)
    d = (torch.from_numpy(r["params"]) - want)
    assert d.norm() <= 1e-4 * want.norm(), float(d.norm() / want.norm())


def test_slabbed_exchange_equals_one_all_reduce_bit_for_bit():
    """The backward-staged path (4 partial backward passes, 4 asynchronous slab all-reduces) against ONE all-reduce of the
    whole flat gradient after a plain backward: same operands per element, hence identical parameters on world_size 2."""
    a = _run("flat", same_data=False, staged=True)
    b = _run("flat", same_data=False, staged=False)
    assert a["replicas_equal"] and b["replicas_equal"] and a["opt_steps"] == b["opt_steps"] == 2
    assert (a["params"] == b["params"]).all()
    assert a["loss"] == b["loss"]
