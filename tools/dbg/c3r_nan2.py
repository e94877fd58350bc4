"""Run hybrid (DP, bf16) to the first non-finite gradient, then replay that step from the saved state in hybrid and flat."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, WORKLOADS, build_dp_policy, clone_batch, make_dp_batch

dev = torch.device("cuda:0")
wl = WORKLOADS["C3R"]
B = wl["batch"]
batches = [make_dp_batch(B, wl["n_points"], seed=1000 + 97 * i, ragged=True, device=dev) for i in range(4)]
gen = torch.Generator().manual_seed(3)
noises = [torch.randn(B, 16, 7, generator=gen).to(dev) for _ in range(64)]
tss = [torch.randint(0, 100, (B,), generator=gen).to(dev) for _ in range(64)]

def make(mode):
    torch.manual_seed(1000)
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    return BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode=mode, optim=dict(DP_OPTIM))

def badnames(tr):
    opt = tr.optimizer
    index = {id(p): k for k, p in enumerate(opt.params)}
    return [n for n, p in tr.policy.named_parameters() if id(p) in index and not torch.isfinite(opt.g_views[index[id(p)]]).all()]

tr = make("hybrid")
fail = None
for i in range(48):
    sd = copy.deepcopy(tr.state_dict())
    b = clone_batch(batches[i % 4]); b["noise"], b["timesteps"] = noises[i], tss[i]
    out = tr.training_step(b)
    if not torch.isfinite(tr.optimizer.flat_g).all():
        fail = i
        print("hybrid: non-finite grads at step", i, badnames(tr), "loss", out["loss"].item())
        break
print("fail step", fail)
if fail is not None:
    for mode in ("hybrid", "flat", "hybrid"):
        t2 = make(mode)
        if mode == "hybrid":  # capture first (its warm-up must not touch the loaded state)
            b = clone_batch(batches[0]); b["noise"], b["timesteps"] = noises[0], tss[0]
            t2.training_step(b)
        t2.load_state_dict(copy.deepcopy(sd))
        b = clone_batch(batches[fail % 4]); b["noise"], b["timesteps"] = noises[fail], tss[fail]
        out = t2.training_step(b)
        print("replay in", mode, ": finite grads:", bool(torch.isfinite(t2.optimizer.flat_g).all()), badnames(t2), "loss", out["loss"].item())
