"""Is a C2 training run reproducible (a) inside one process -- two trainers built from the same seeds -- and (b) across processes?
Prints a checksum of the flat parameter buffer after N steps; run it twice to compare (b)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(tag):
    torch.manual_seed(1000)
    policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(policy, total_steps=200, precision=os.environ.get("PREC", "bf16"), device=dev, mode=os.environ.get("MODE", "graph"),
                   optim=dict(accumulate_grad_batches=1), external_sampling=os.environ.get("EXT", "1") == "1")
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(4)]
    sums = []
    for i in range(steps):
        tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4] if os.environ.get("PREFETCH", "1") == "1" else None)
        if os.environ.get("SYNC", "0") == "1":
            torch.cuda.synchronize()
        if i in (0, 1, 2, 3, 4, 5, 6, 8, steps - 1):
            torch.cuda.synchronize()
            sums.append((i, hashlib.sha1(tr.optimizer.flat_p.detach().cpu().numpy().tobytes()).hexdigest()[:12],
                         hashlib.sha1(tr.optimizer.flat_g.detach().cpu().numpy().tobytes()).hexdigest()[:12]))
    print(tag, sums, flush=True)
    return sums


a = run("run A")
b = run("run B")
print("same inside the process:", a == b)
