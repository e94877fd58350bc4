"""A soak with a NEW synthetic batch every step (the bench cycles four): graph mode (equal-size clouds) and hybrid mode (ragged), prefetch on,
no host synchronisation except every 200 steps, where it checks: loss finite, allocator footprint not growing, and this step's static FPS /
kNN indices equal to the CPU oracle's for the same clouds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pointops_cpu  # noqa: E402  (test infrastructure: the checker)
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for mode, ragged, wname in (("graph", False, "C2"), ("hybrid", True, "C2")):
    wl = WORKLOADS[wname]
    torch.manual_seed(1000)
    policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(policy, total_steps=steps + 10, precision="bf16", device=dev, mode=mode, optim=dict(accumulate_grad_batches=1))
    nxt = make_act_batch(wl["batch"], wl["n_points"], seed=5000, ragged=ragged, device=dev)
    mem0 = None
    for i in range(steps):
        cur, nxt = nxt, make_act_batch(wl["batch"], wl["n_points"], seed=5001 + i, ragged=ragged, device=dev)
        st = tr.training_step(clone_batch(cur), prefetch=nxt)
        if i % 200 == 199 or i == steps - 1:
            torch.cuda.synchronize()
            loss = float(st["loss"])
            mem = torch.cuda.memory_allocated() >> 20
            mem0 = mem if mem0 is None else mem0
            msg = f"{mode} step {i + 1}: loss {loss:.4f} finite {loss == loss and abs(loss) < 1e6}  allocated {mem} MiB (first check {mem0})"
            if mode == "graph":  # the static index buffers hold THIS step's sampling
                pre = policy._static_pre["pre"]
                pc = cur["pcds"]
                p, o = pc["coord"].cpu(), pc["offset"].cpu()
                n_o = policy._new_offsets(pc["offset"]).cpu()
                want = pointops_cpu.farthest_point_sampling(p, o, n_o)
                wk, _ = pointops_cpu.knn_query(policy.pcd_nsample, p, o, p[want.long()], n_o)
                msg += f"  FPS == oracle {torch.equal(pre['idx'].cpu(), want)}  kNN == oracle {torch.equal(pre['knn_idx'].cpu(), wk)}"
            print(msg, flush=True)
    print(f"{mode}: prefetch entries left over: {len(policy.__dict__.get('_prefetched', {}))}", flush=True)
    del tr, policy
    torch.cuda.empty_cache()
