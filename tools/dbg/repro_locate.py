"""Localise run-to-run differences of the bf16 C2 step: the SAME micro-batch, weights and dropout seed evaluated REPS times in flat
mode (eager launches, no optimizer step in between); prints which repetitions differ from the first and in which parameters'
gradients (by name), plus loss terms."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS[os.environ.get("WL", "C2")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
torch.manual_seed(1000)
policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(policy, total_steps=200, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
batch = make_act_batch(wl["batch"], wl["n_points"], seed=1000, ragged=False, device=dev)
opt = tr.optimizer
names = {id(p): n for n, p in policy.named_parameters()}
ref = None
for r in range(reps):
    tr._fused_ctx.set_step(7)
    tr._fused_ctx.site = 0
    opt.flat_g.zero_()
    for k in range(len(opt.params)):
        opt._stash[k] = None
        opt.params[k].grad = None
    stats = tr._forward_backward(clone_batch(batch), first=True)
    torch.cuda.synchronize()
    g = opt.flat_g.detach().clone()
    st = stats.detach().clone() if torch.is_tensor(stats) else torch.tensor([float(v) for v in stats.values()])
    if ref is None:
        ref, ref_st = g, st
        continue
    if not torch.equal(g, ref) or not torch.equal(st, ref_st):
        bad = []
        for k, p in enumerate(opt.params):
            o = opt.offsets[k]
            a, b = g[o:o + p.numel()], ref[o:o + p.numel()]
            if not torch.equal(a, b):
                bad.append((names[id(p)], int((a != b).sum()), float((a - b).abs().max()), float(b.abs().max())))
        print(f"rep {r}: differs; stats equal: {torch.equal(st, ref_st)}; {len(bad)} parameters differ; first 12:", bad[:12], flush=True)
    else:
        print(f"rep {r}: identical", flush=True)
