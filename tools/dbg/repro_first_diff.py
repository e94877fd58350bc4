"""Two runs of N flat-mode bf16 steps from the same seeds inside one process (module-level batching state reset in between, a device
synchronisation after every step): the first step at which loss / gradients / parameters differ, and in which parameters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402
from pointcloudmatters_amd.policy import deferred  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = os.environ.get("MODE", "flat")


def run():
    for d in (deferred._EXPECT, deferred._TAKE_EXPECT, deferred._SEEN, deferred._TAKEN):
        d.clear()
    torch.manual_seed(1000)
    policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(policy, total_steps=200, precision="bf16", device=dev, mode=mode, optim=dict(accumulate_grad_batches=1))
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(4)]
    rec = []
    for i in range(steps):
        st = tr.training_step(clone_batch(batches[i % 4]))
        torch.cuda.synchronize()
        rec.append((float(st["loss"]), tr.optimizer.flat_g.detach().clone(), tr.optimizer.flat_p.detach().clone()))
    names = {}
    for n, p in policy.named_parameters():
        names[id(p)] = n
    layout = [(names[id(p)], o, p.numel()) for p, o in zip(tr.optimizer.params, tr.optimizer.offsets)]
    return rec, layout


a, layout = run()
b, _ = run()
for i, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(a, b)):
    eg, ep = torch.equal(ga, gb), torch.equal(pa, pb)
    print(f"step {i}: loss {la:.6f} / {lb:.6f}  grads equal {eg}  params equal {ep}")
    if not eg:
        bad = [(n, int((ga[o:o + k] != gb[o:o + k]).sum()), float((ga[o:o + k] - gb[o:o + k]).abs().max())) for n, o, k in layout
               if not torch.equal(ga[o:o + k], gb[o:o + k])]
        print("   first differing step: ", len(bad), "parameters differ:", bad[:10])
        break
