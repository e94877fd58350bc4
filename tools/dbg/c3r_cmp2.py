"""hybrid vs flat with the real lr: first parameter (or buffer) whose trajectory diverges."""
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
N = 8
for mode in ("flat", "hybrid"):
    torch.manual_seed(1000)
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode=mode, optim=dict(DP_OPTIM))
    snaps = []
    for i in range(N):
        b = clone_batch(batches[i % 3]); b["noise"], b["timesteps"] = noise, tsteps
        out = tr.training_step(b)
        opt = tr.optimizer
        index = {id(p): k for k, p in enumerate(opt.params)}
        snaps.append(({n: v.detach().clone() for n, v in pol.state_dict().items()},
                      {n: opt.g_views[index[id(p)]].detach().clone() for n, p in pol.named_parameters() if id(p) in index},
                      out["loss"].item(), opt.flat_p_bf16.detach().clone(), float(opt.grad_norm[0])))
    res[mode] = snaps
for i in range(N):
    sa, ga, la, ma, na = res["flat"][i]; sb, gb, lb, mb, nb = res["hybrid"][i]
    print(f"step {i}: loss flat {la:.5f} hybrid {lb:.5f}  gnorm {na:.4f} {nb:.4f} mirror diff {(ma.float()-mb.float()).abs().max().item():.3e}")
    for what, A, Bd in (("grad", ga, gb), ("state", sa, sb)):
        worst = []
        for n in A:
            if not A[n].dtype.is_floating_point: 
                if not torch.equal(A[n], Bd[n]): print("    int state differs", n, A[n], Bd[n])
                continue
            d = (A[n].float() - Bd[n].float()).norm().item(); r = A[n].float().norm().item()
            worst.append((d / (r + 1e-12), n))
        worst.sort(reverse=True)
        print("   ", what, ["%.2e %s" % w for w in worst[:5]])
