"""Per-parameter gradient comparison hybrid vs flat for the DP workload (bf16), identical noise / timesteps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, WORKLOADS, build_dp_policy, clone_batch, make_dp_batch

dev = torch.device("cuda:0")
wl = WORKLOADS["C3R"]
B = wl["batch"]
g = torch.Generator().manual_seed(3)
noise = torch.randn(B, 16, 7, generator=g).to(dev)
tsteps = torch.randint(0, 100, (B,), generator=g).to(dev)
batches = [make_dp_batch(B, wl["n_points"], seed=1000 + 97 * i, ragged=True, device=dev) for i in range(3)]
res = {}
for mode in ("flat", "hybrid"):
    torch.manual_seed(1000)
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode=mode, optim=dict(DP_OPTIM, lr=1e-9))
    gs = []
    for i in range(4):
        b = clone_batch(batches[i % 3]); b["noise"], b["timesteps"] = noise, tsteps
        out = tr.training_step(b)
        opt = tr.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        gs.append(({n: opt.g_views[index[id(p)]].detach().clone() for n, p in pol.named_parameters() if id(p) in index}, out["loss"].item()))
    res[mode] = gs
for i in range(4):
    ga, la = res["flat"][i]; gb, lb = res["hybrid"][i]
    worst = []
    for n in ga:
        d = (ga[n] - gb[n]).norm().item(); r = ga[n].norm().item()
        worst.append((d / (r + 1e-12), n, r, gb[n].norm().item()))
    worst.sort(reverse=True)
    print(f"step {i}: loss flat {la:.5f} hybrid {lb:.5f}")
    for w in worst[:8]:
        print("    rel %.3e  %s  |flat| %.3e |hyb| %.3e" % w)
