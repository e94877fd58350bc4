"""CPU: the product's DATA-PARALLEL trainer on two ranks (gloo), each rank running the kernel sources on the host wave64 model
(tests/wavesim), against ONE process that trains on the concatenated batch -- the CPU twin of tests/test_hybrid_two_ranks_gpu.py, whose
helpers it reuses.  mode="flat" (the model has no hipGraphs): the flat AdamW kernel, gradient slabs all-reduced per backward stage,
synchronised BatchNorm INSIDE the fused kernels (csrc/bnrelu.hip / sa_fused.hip local sums -> pack -> all_gather -> combine kernel).
With SyncBN the two runs are the same optimisation problem: replicas bit-identical, losses and updates equal to the single process's.
What it cannot show: RCCL, xGMI, overlap -- the collectives here are gloo's."""
import os
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import test_hybrid_two_ranks_gpu as H
from tests.wavesim import build as _build

pytestmark = pytest.mark.skipif(not os.path.exists(_build.CLANG), reason="needs the ROCm clang++ as host compiler: " + _build.CLANG)


def _worker(rank, world, port, q, kind):
    from tests.wavesim.backend import simulated_device

    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with simulated_device(claim_cuda=True) as dev:
            log = H._CollectiveLog()
            log.install()
            tr, losses = H._train(dev, H._rank_batches(rank, dev, kind), H._eps(rank, kind), distributed=True, kind=kind,
                                  tsteps=H._timesteps(rank), mode="flat")
            assert tr.mode == "flat" and tr.distributed and tr.sync_batchnorm
            slabs = [c for c in log.calls if c[0] == "all_reduce" and c[1] >= 1024]
            q.put({"losses%d" % rank: np.asarray(losses), "params%d" % rank: H._named(tr),
                   "rm%d" % rank: H._bn_of(tr, kind).running_mean.detach().cpu().numpy(),
                   "ncoll%d" % rank: np.asarray([len(log.calls), len(slabs)]), "stages%d" % rank: np.asarray([len(tr._stages)])})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["act", "dp"])
def test_two_ranks_on_the_model_equal_one_process_on_the_whole_batch(kind):
    from tests.wavesim.backend import simulated_device

    _build.build()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = H._free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(9000):
        while not q.empty():
            got.update(q.get())
        if len(got) >= 10 or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    for p in procs:
        p.join(300)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert len(got) == 10
    assert got["ncoll0"].tolist() == got["ncoll1"].tolist() and got["ncoll0"][1] >= H.STEPS  # same collectives; >= one slab exchange per step
    for n in got["params0"]:  # replicas stay identical
        np.testing.assert_array_equal(got["params0"][n], got["params1"][n], err_msg=n)
    np.testing.assert_array_equal(got["rm0"], got["rm1"])
    with simulated_device(claim_cuda=True) as dev:
        b0, b1 = H._rank_batches(0, dev, kind), H._rank_batches(1, dev, kind)
        whole = [H._concat(x, y) for x, y in zip(b0, b1)]
        eps = torch.cat([H._eps(0, kind), H._eps(1, kind)], dim=1)
        tr, losses = H._train(dev, whole, eps, distributed=False, kind=kind, tsteps=torch.cat([H._timesteps(0), H._timesteps(1)], dim=1),
                              mode="flat")
        ref = H._named(tr)
        index = {id(p): k for k, p in enumerate(tr.optimizer.params)}
        gmax = {n: float(tr.optimizer.g_views[index[id(p)]].abs().max()) if id(p) in index else 0.0 for n, p in tr.policy.named_parameters()}
        rm = H._bn_of(tr, kind).running_mean.detach().cpu().numpy()
    assert (got["losses0"] + got["losses1"]) / 2 == pytest.approx(np.asarray(losses), rel=2e-4)
    init = {n: p.detach().float().numpy() for n, p in H._build(kind)[0].named_parameters()}
    gscale, worst = max(gmax.values()), (0.0, None)
    for n in ref:
        moved = float(np.linalg.norm(ref[n] - init[n]))
        if moved < 1e-6 or gmax[n] < 1e-5 * gscale:  # frozen parameters; biases in front of a BatchNorm (see the GPU twin)
            continue
        worst = max(worst, (float(np.linalg.norm(got["params0"][n] - ref[n])) / moved, n))
    assert worst[0] <= 0.05, worst
    # running statistics after three Adam steps: the weights in front of the BatchNorm have drifted by the noise described in the GPU twin
    # (an element whose gradient is ~0 moves by +-lr); the host GEMMs make that a little larger than on the device
    np.testing.assert_allclose(got["rm0"], rm, rtol=2e-3, atol=1e-4)
