"""Where does the transient non-finite gradient of hybrid mode come from?  Check flat_g right after the graph replay, after the
eager tokenizer backward, and re-replay the graph on the same inputs when it happens."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudmatters_amd.bc import DP_OPTIM, BCTrainer, WORKLOADS, build_dp_policy, clone_batch, make_dp_batch
from pointcloudmatters_amd.bc import trainer as T

dev = torch.device("cuda:0")
wl = WORKLOADS["C3R"]
B = wl["batch"]
batches = [make_dp_batch(B, wl["n_points"], seed=1000 + 97 * i, ragged=True, device=dev) for i in range(4)]
torch.manual_seed(1000)
pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="hybrid", optim=dict(DP_OPTIM))
sync = "--sync" in sys.argv

orig_replay = torch.cuda.CUDAGraph.replay
state = {}
def replay(self):
    orig_replay(self)
    torch.cuda.synchronize()
    g = tr.optimizer.flat_g
    if not torch.isfinite(g).all():
        bad = (~torch.isfinite(g)).nonzero().flatten()
        print("  after replay: nonfinite at", bad[:8].tolist(), "count", bad.numel(), g[bad[:8]].tolist(), flush=True)
        opt = tr.optimizer
        names = {id(p): n for n, p in pol.named_parameters()}
        for o in bad[:8].tolist():
            for k, (pp, off) in enumerate(zip(opt.params, opt.offsets)):
                if off <= o < off + pp.numel():
                    print("     ", o, names[id(pp)], "k", k, "in subset_a" if k in tr._subset_a_set else "subset_b", "shadow" if opt.shadow[k] is not None else "master", "idx in param", o - off, "numel", pp.numel())
                    break
            else:
                print("     ", o, "in padding")
        orig_replay(self); torch.cuda.synchronize()
        print("  re-replay finite:", bool(torch.isfinite(g).all()), flush=True)
        # the same stage on the same static inputs, eagerly
        from pointcloudmatters_amd.policy import fused_ops
        opt = tr.optimizer
        for rep in range(3):
            g.zero_()
            with fused_ops.activate(tr._fused_ctx), tr._autocast():
                data = pol.hybrid_merge(clone_batch(tr._static_batch), (tr._static_tokens,) + tr._static_extra)
                out = tr._call_policy(data)
            tr._static_tokens.grad = None
            out["loss"].backward()
            opt.collect(first=True, subset=tr._subset_b)
            torch.cuda.synchronize()
            print("  eager stage B finite:", bool(torch.isfinite(g).all()), "loss", out["loss"].item(), flush=True)
        print("  static tokens finite", bool(torch.isfinite(tr._static_tokens).all()), "abs max", tr._static_tokens.abs().max().item())
        for rep in range(3):
            orig_replay(self); torch.cuda.synchronize()
            print("  re-replay finite:", bool(torch.isfinite(g).all()), "nbad", int((~torch.isfinite(g)).sum()), flush=True)
        state["bad"] = True
torch.cuda.CUDAGraph.replay = replay
for i in range(60):
    out = tr.training_step(clone_batch(batches[i % 4]))
    if sync: torch.cuda.synchronize()
    if state.get("bad") or not torch.isfinite(tr.optimizer.flat_g).all():
        print("step", i, "bad; after-replay bad:", state.get("bad"))
        break
else:
    print("no failure in 60 steps (sync=%s)" % sync)
