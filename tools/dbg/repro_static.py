"""Graph-mode bf16 C2 steps with NO host synchronisation: per step, fingerprints of (1) the prefetched FPS indices at the moment the trainer
loads them, (2) the static index buffer right behind that copy, (3) the same buffer after the replay + optimizer.  All snapshots are
device-side clones on the current stream, hashed after the run."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402
from pointcloudmatters_amd.policy import sa_layer  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
torch.manual_seed(1000)
policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(policy, total_steps=200, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(4)]
log = []
cur = {}
orig = sa_layer.load_static


def load_static(owner, pre):
    if pre.get("event") is not None:
        torch.cuda.current_stream().wait_event(pre["event"])
    cur["pre"] = pre["idx"].clone()
    orig(owner, pre)
    cur["after_copy"] = owner._static_pre["pre"]["idx"].clone()


sa_layer.load_static = load_static
sa_layer.set_abstraction.load_static = load_static if hasattr(sa_layer.set_abstraction, "load_static") else None
for i in range(steps):
    cur.clear()
    tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4])
    cur["after_step"] = policy._static_pre["pre"]["idx"].clone()
    log.append(dict(cur))
torch.cuda.synchronize()


def h(t):
    return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:8] if t is not None else "--------"


good = {i: log[i]["pre"] for i in range(4)}
M = wl["pcd_npoints"]
for i, e in enumerate(log):
    a, b = e["pre"].view(-1, M), good[i % 4].view(-1, M)
    if not torch.equal(a, b):
        rows = [(c, int((a[c] != b[c]).nonzero()[0]), int((a[c] != b[c]).sum())) for c in range(a.shape[0]) if not torch.equal(a[c], b[c])]
        print("step", i, ": clouds whose picks differ (cloud, first differing pick, number of differing picks):", rows,
              " values at the first:", [(int(a[c][f]), int(b[c][f])) for c, f, _ in rows][:4])
ref = {}
for i, e in enumerate(log):
    hs = [h(e.get(k)) for k in ("pre", "after_copy", "after_step")]
    ref.setdefault(i % 4, hs[2] if i < 4 else None)
    print(i, "batch", i % 4, " prefetched", hs[0], " static after copy", hs[1], " static after step", hs[2],
          "" if len(set(hs)) == 1 else "   <-- differ", flush=True)
