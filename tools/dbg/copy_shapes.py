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
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.training_step(clone_batch(batches[0]))
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks or e.name not in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::fill_", "aten::mul", "aten::div"):
        continue
    key = (e.name, str(e.input_shapes)[:120], ks[0].name.split("(")[0][-60:])
    agg[key][0] += len(ks); agg[key][1] += sum(k.duration for k in ks)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%4d x %7.1f us  %-10s %-120s %s" % (v[0], v[1], k[0], k[1], k[2]))
