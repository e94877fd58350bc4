"""C2 graph step with / without the CVAE-encoder stream fork inside the captured graph (same box, alternating)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
wl = WORKLOADS[name]
for rep in range(2):
    for fork in (True, False):
        torch.manual_seed(1000)
        pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
        pol.fork_cvae = fork
        tr = BCTrainer(pol, total_steps=1000, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
        batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
        for i in range(8):
            tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
        torch.cuda.synchronize()
        print(name, "fork_cvae", fork, "ms/step %.3f" % ((time.perf_counter() - t0) / 40 * 1e3), "loss %.4f" % tr.metrics()["train/loss"], flush=True)
        del tr, pol
