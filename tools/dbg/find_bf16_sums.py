"""Which framework bf16 reductions are left in the ACT training step (flat mode = the same ops the captured graph holds)?  Lists aten::sum
calls with a bf16 input: shape, and the Python frames that issued them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import (DP_OPTIM, WORKLOADS, BCTrainer, build_act_policy, build_dp_policy, clone_batch, make_act_batch,  # noqa: E402
                                          make_dp_batch)

dev = torch.device("cuda", 0)
WL = os.environ.get("WL", "C2")
wl = WORKLOADS[WL]
torch.manual_seed(1000)
if wl["policy"] == "dp":
    policy = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(policy, total_steps=50, precision="bf16", device=dev, mode="flat", optim=dict(DP_OPTIM))
    batch = make_dp_batch(wl["batch"], wl["n_points"], seed=1000, ragged=False, device=dev)
else:
    policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(policy, total_steps=50, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
    batch = make_act_batch(wl["batch"], wl["n_points"], seed=1000, ragged=False, device=dev)
for _ in range(3):
    tr.training_step(clone_batch(batch))
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.training_step(clone_batch(batch))
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.name in ("aten::sum", "aten::mean", "aten::sum_to_size") and ev.input_shapes:
        key = (ev.name, str(ev.input_shapes[:2]), tuple(str(f) for f in (ev.stack or [])[:6] if "pointcloudmatters" in str(f) or "torch/nn" in str(f)))
        seen[key] = seen.get(key, 0) + 1
for (name, shapes, stack), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, name, shapes)
    for f in stack[:4]:
        print("      ", f)
