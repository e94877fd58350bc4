"""C2 graph step with PyTorch TunableOp picking the GEMM solution per shape (tuned during the warm-up runs)."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
tun = len(sys.argv) > 1 and sys.argv[1] == "tune"
if tun:
    t = torch.cuda.tunable
    t.enable(True); t.tuning_enable(True)
    t.set_filename("/root/repo/gpurun_out/tunable_c2.csv")
    t.set_max_tuning_duration(15); t.set_max_tuning_iterations(20)
torch.manual_seed(1000)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
t0 = time.perf_counter()
tr.training_step(clone_batch(batches[0]))
torch.cuda.synchronize()
print("first step (warm-up + tuning + capture) %.1f s" % (time.perf_counter() - t0), flush=True)
for i in range(6):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
print("tunable", tun, "ms/step %.3f" % ((time.perf_counter() - t0) / 40 * 1e3), {k: round(float(v), 5) for k, v in tr.metrics().items() if "loss" in k})
if tun:
    torch.cuda.tunable.write_file() if hasattr(torch.cuda.tunable, "write_file") else None
    print("results", len(torch.cuda.tunable.get_results()))
