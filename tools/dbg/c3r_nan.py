"""Bisect the C3R NaN: run the DP workload in several configurations, report the first non-finite step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, WORKLOADS, build_dp_policy, clone_batch, make_dp_batch

dev = torch.device("cuda:0")
wl = WORKLOADS["C3R"]

def run(mode, sa_impl, ragged, precision="bf16", steps=40, prefetch=True):
    torch.manual_seed(1000)
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl=sa_impl).to(dev)
    tr = BCTrainer(pol, total_steps=100, precision=precision, device=dev, mode=mode, optim=dict(DP_OPTIM))
    batches = [make_dp_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=ragged, device=dev) for i in range(4)]
    bad = None
    for i in range(steps):
        out = tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4] if prefetch else None)
        l = out["loss"].item()
        g = tr.optimizer.flat_g
        gfin = bool(torch.isfinite(g).all().item())
        pfin = bool(torch.isfinite(tr.optimizer.flat_p).all().item())
        if not (l == l) or not gfin or not pfin:
            bad = (i, l, gfin, pfin, float(tr.optimizer.grad_norm[0]))
            # which parameter gradients are non-finite
            names = []
            index = {id(p): k for k, p in enumerate(tr.optimizer.params)}
            for n, p in pol.named_parameters():
                if id(p) in index and not torch.isfinite(tr.optimizer.g_views[index[id(p)]]).all():
                    names.append(n)
            print("   nonfinite grads:", names[:12], len(names))
            break
    print(f"mode={mode} sa={sa_impl} ragged={ragged} prec={precision} prefetch={prefetch}: first bad = {bad}; last loss {l:.4f} gnorm {float(tr.optimizer.grad_norm[0]):.3f}", flush=True)

for cfg in [("hybrid", "fused", True), ("flat", "fused", True), ("hybrid", "fused", False), ("graph", "fused", False), ("flat", "torch", True), ("hybrid", "fused", True, "fp32")]:
    try:
        run(*cfg)
    except Exception as e:
        print(cfg, "EXC", type(e).__name__, e, flush=True)
run("hybrid", "fused", True, prefetch=False)
