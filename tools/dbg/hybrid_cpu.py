"""Is hybrid mode CPU-bound?  Enqueue time of N steps (no sync) vs completion time."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
wl = WORKLOADS[name]
torch.manual_seed(1000)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=1000, precision="bf16", device=dev, mode=mode, optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=mode == "hybrid", device=dev) for i in range(4)]
for i in range(10):
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for i in range(N):
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(name, mode, "enqueue %.3f ms/step, complete %.3f ms/step (CPU-bound if equal)" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
