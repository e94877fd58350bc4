"""The N > 1 program on ONE GPU: a one-rank RCCL group (PCM_DP_SINGLE_RANK=1) makes the trainer issue every collective of the
data-parallel path for real -- 6 + 6 synchronised-BatchNorm exchanges, 4 gradient slabs -- so that what is measured is the
program's own overhead (host pacing, graph launches, stream hand-overs to RCCL and back), not link latency.
Usage: python tools/dbg/dp_single_rank.py [C2|C4]"""
import os
import sys
import time

os.environ.setdefault("PCM_DP_SINGLE_RANK", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run(wl_name, mode, distributed, steps=60, warmup=12):
    from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch

    wl = WORKLOADS[wl_name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=1000, precision=wl["dtype"], device=dev, mode=mode, distributed=distributed,
                   optim=dict(accumulate_grad_batches=1))
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(4)]

    def step(i):
        tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    extra = ""
    if mode == "graph" and getattr(tr, "segmented", False):
        extra = " chain: %d graphs + %d collectives" % (tr._graph[0].n_graphs, tr._graph[0].n_calls)
    print(f"{wl_name} mode={mode:6s} collectives={'on ' if distributed else 'off'} {ms:7.3f} ms/step  sync_bn={tr.sync_batchnorm}{extra}", flush=True)
    del tr, pol
    torch.cuda.empty_cache()


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    only = sys.argv[2] if len(sys.argv) > 2 else None  # "graph" | "hybrid": one configuration, few steps (for rocprofv3)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    if only:
        run(wl, only, True, steps=12, warmup=6)
    else:
        run(wl, "graph", False)   # the N = 1 program
        run(wl, "hybrid", True)   # round-3 N > 1 program
        run(wl, "graph", True)    # round-4 N > 1 program
    dist.destroy_process_group()


main()
