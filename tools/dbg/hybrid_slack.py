"""Is hybrid mode host-paced?  Add an artificial host delay per step: a GPU-bound loop absorbs it, a host-paced one does not."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
for mode in ("hybrid", "graph"):
    torch.manual_seed(1000)
    pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=1000, precision="bf16", device=dev, mode=mode, optim=dict(accumulate_grad_batches=1))
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=mode == "hybrid", device=dev) for i in range(4)]
    for i in range(10):
        tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
    torch.cuda.synchronize()
    for delay in (0.0, 0.001, 0.002, 0.004):
        t0 = time.perf_counter()
        for i in range(40):
            tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
            if delay:
                time.sleep(delay)
        torch.cuda.synchronize()
        print(mode, "host delay %.0f ms -> %.3f ms/step" % (delay * 1e3, (time.perf_counter() - t0) / 40 * 1e3), flush=True)
