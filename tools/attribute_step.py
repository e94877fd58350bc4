"""Which line of the policy launches what?  Runs the C2 step in mode="flat" (same kernels as the replayed graph, but launched
from Python) under torch.profiler with stacks and attributes every device kernel to the innermost frame inside
pointcloudmatters_amd/ (plus the aten op), for forward and backward (backward ops carry the stack of their autograd node's
creation only partially, so they are grouped by aten op + autograd node name)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch

dev = torch.device("cuda:0")
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
torch.manual_seed(1000)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
for i in range(4):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    for i in range(STEPS):
        tr.training_step(clone_batch(batches[i % 4]))
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    # only leaf-most CPU ops own kernels in this list
    frame = None
    for fr in (e.stack or []):
        if "pointcloudmatters_amd" in fr and "site-packages" not in fr:
            frame = fr.split("pointcloudmatters_amd/")[-1]
            break
    where = frame or ("<autograd/other> " + e.name)
    key = (where, e.name)
    for k in ks:
        agg[key][0] += k.duration
        agg[key][1] += 1
        agg[key][2][k.name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:50]] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("total %.0f us/step, %d launches/step" % (tot / STEPS, sum(v[1] for v in agg.values()) / STEPS))
for (where, op), (us, n, kinds) in rows[:90]:
    print("%7.1f us %5.1f x  %-70s %-28s %s" % (us / STEPS, n / STEPS, where[:70], op[:28], ", ".join("%s*%d" % (k[:34], c // STEPS) for k, c in kinds.most_common(2))))
