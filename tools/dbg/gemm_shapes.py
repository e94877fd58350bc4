"""GEMM inventory of one C2 step (flat mode): aten op + input shapes -> launches / device time."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch

dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
torch.manual_seed(1000)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
for i in range(4):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    for i in range(STEPS):
        tr.training_step(clone_batch(batches[i % 4]))
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
other = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    t = sum(k.duration for k in ks)
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::_scaled_mm"):
        key = (e.name, str(e.input_shapes))
        agg[key][0] += t; agg[key][1] += len(ks)
        for k in ks: agg[key][2][k.name[:60]] += 1
    else:
        other[e.name][0] += t; other[e.name][1] += len(ks)
tot = sum(v[0] for v in agg.values())
print("GEMM total %.1f us/step, %d launches/step" % (tot / STEPS, sum(v[1] for v in agg.values()) / STEPS))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%7.1f us/step %5.1f launches  avg %5.1f  %s %s" % (v[0] / STEPS, v[1] / STEPS, v[0] / v[1], k[0], k[1]))
print("---- non-GEMM ops")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%7.1f us/step %5.1f launches  avg %5.1f  %s" % (v[0] / STEPS, v[1] / STEPS, v[0] / v[1], k))
