"""Is the training step host-bound?  Host time per training_step() call (no synchronisation inside the loop) against the device
time per step, and the device's idle time per step from a torch.profiler kernel trace.  python tools/dbg/host_vs_gpu.py [WORKLOAD] [MODE]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else None
wl = WORKLOADS[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
mode = mode or ("hybrid" if wl.get("ragged") else "graph")
kw = {}
if "PCM_DBG_STAGED" in os.environ:
    kw["staged"] = bool(int(os.environ["PCM_DBG_STAGED"]))
if "PCM_DBG_SPLIT" in os.environ:
    kw["overlap_rest_backward"] = bool(int(os.environ["PCM_DBG_SPLIT"]))
tr = BCTrainer(pol, total_steps=10000, precision="bf16", device=dev, mode=mode, optim=dict(accumulate_grad_batches=1), **kw)
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=s, ragged=wl.get("ragged", False), device=dev) for s in range(4)]
for i in range(8):
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
torch.cuda.synchronize()
N = 30
host = []
t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter()
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(f"{name} {mode}: wall {t_all / N * 1e3:.3f} ms/step; host enqueue {t_enq / N * 1e3:.3f} ms/step (median call {host[N // 2] * 1e3:.3f} ms); "
      f"host is {'AHEAD' if t_enq < 0.9 * t_all else 'THE BOTTLENECK'}")
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for i in range(6):
        tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
    torch.cuda.synchronize()
ev = sorted((e.time_range.start, e.time_range.end) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start)
cs, ce = ev[0]
busy, gaps = 0, []
for s_, e_ in ev[1:]:
    if s_ > ce:
        gaps.append(s_ - ce)
        busy += ce - cs
        cs, ce = s_, e_
    else:
        ce = max(ce, e_)
busy += ce - cs
span = ev[-1][1] - ev[0][0]
big = sorted(gaps, reverse=True)[:12]
if "--around" in sys.argv:  # what runs before / after the largest gap
    evn = sorted((e.time_range.start, e.time_range.end, e.name) for e in prof.events()
                 if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start)
    ends, best, cur_end = [], None, evn[0][1]
    for i in range(1, len(evn)):
        if evn[i][0] > cur_end and (best is None or evn[i][0] - cur_end > best[0]) and i > len(evn) // 2:
            best = (evn[i][0] - cur_end, i)
        cur_end = max(cur_end, evn[i][1])
    g, i = best
    print("largest gap in the second half: %.1f us; kernels around it:" % g)
    for a, b, n in evn[i - 6:i + 8]:
        print("   %10.1f  %7.1f us  %s" % (a - evn[i][0], b - a, n[:90]))
print(f"profiled 6 steps: span {span / 6e3:.3f} ms/step, device busy (union over streams) {busy / 6e3:.3f} ms/step, idle {(span - busy) / 6e3:.3f} ms/step; "
      f"largest gaps (us): {[round(g, 1) for g in big]}")
