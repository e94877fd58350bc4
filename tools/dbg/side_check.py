"""C2 graph-mode step with and without side-stream weight gradients: ms/step and loss trajectories."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
from pointcloudmatters_amd.policy import rows_linear
dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
res = {}
for side, mr in ((False, 0), (True, 2048), (True, 4000), (True, 8000)):
    rows_linear.SIDE.min_rows = mr
    torch.manual_seed(1000)
    pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1), side_weight_grads=side)
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
    losses = []
    for i in range(6):
        tr.training_step(clone_batch(batches[i % 4]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        tr.training_step(clone_batch(batches[i % 4]))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    m = tr.metrics()
    res[side] = (ms, m)
    print("side", side, mr, "ms/step %.3f" % ms, {k: round(float(v), 5) for k, v in m.items() if "loss" in k or "norm" in k})
