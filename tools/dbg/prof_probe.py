import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
torch.manual_seed(1000)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
for i in range(6):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(5):
        tr.training_step(clone_batch(batches[i % 4]))
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None: dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0:
        rows.append((dt, e.count, e.key, str(getattr(e, "device_type", ""))))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total device us", tot, "events", len(rows), "launches", sum(r[1] for r in rows))
for r in rows[:70]:
    print("%9.1f us %5d  %s  %s" % (r[0], r[1], r[2][:130], r[3]))
