"""Per-step fingerprints of a graph-mode bf16 C2 run WITHOUT any host synchronisation during the run (device-side clones, hashed at the
end): gradients, parameters, the static sampling buffers the replayed graph read, the loss.  Run it twice and diff the outputs."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(1000)
policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(policy, total_steps=200, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(4)]
# experiment switches: a device synchronisation at one point of every step
WHERE = os.environ.get("SYNC_AT", "")
target = tr._sampling_target() or policy
_orig_load = target.load_static_sampling
_orig_pref = tr.prefetch_sampling


_keep = []


def _load(pre):
    if WHERE == "before_load":
        torch.cuda.synchronize()
    if WHERE == "hold":      # keep every sampling result alive: the allocator can never hand its memory out again
        _keep.append(pre)
    if WHERE in ("snap_pre", "snap_static"):  # fingerprint of what the copy is ABOUT to read (device-side clone on the main stream, behind the event wait)
        if pre.get("event") is not None:
            torch.cuda.current_stream().wait_event(pre["event"])
        _keep.append([pre["idx"].clone(), pre["knn_idx"].clone()])
    _orig_load(pre)
    if WHERE == "snap_static":  # the static buffers right behind the copy, before the replay
        sp = policy._static_pre["pre"]
        _keep.append([sp["idx"].clone(), sp["knn_idx"].clone()])
    if WHERE == "side_waits":
        policy._side_stream.wait_stream(torch.cuda.current_stream())
    if WHERE == "after_load":
        torch.cuda.synchronize()


def _pref(nb):
    if WHERE == "before_prefetch":
        torch.cuda.synchronize()
    _orig_pref(nb)
    if WHERE == "after_prefetch":
        torch.cuda.synchronize()


target.load_static_sampling = _load
tr.prefetch_sampling = _pref
rec = []
pre_on = os.environ.get("PREFETCH", "1") == "1"
for i in range(steps):
    st = tr.training_step(clone_batch(batches[i % 4]), prefetch=batches[(i + 1) % 4] if pre_on else None)
    pre = policy.__dict__.get("_static_pre")
    snap = [tr.optimizer.flat_g.clone(), tr.optimizer.flat_p.clone(), st["loss"].clone(), tr._fused_ctx.seed.clone(), tr.optimizer.hyper.clone()]
    if pre is not None:
        snap += [t.clone() for t in (pre["pre"]["idx"], pre["pre"]["knn_idx"])] + [t.clone() for t in (pre["pre"].get("istats") or ())]
    rec.append(snap)
torch.cuda.synchronize()


def h(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:8]


good = {}
bad = 0
for i, snap in enumerate(rec):
    hs = [h(t) for t in snap]
    key = tuple(hs[5:])
    if i < 4:
        good[i % 4] = key
    elif good[i % 4] != key:
        bad += 1
    if os.environ.get("VERBOSE"):
        print(i, " ".join(hs), flush=True)
if WHERE in ("snap_pre", "snap_static"):
    print("fingerprints of the prefetched idx / knn at load time, per step:", [(i, h(a), h(b)) for i, (a, b) in enumerate(_keep)][:16])
print("SYNC_AT=%s PREFETCH=%s: steps whose static sampling buffers differ from the first occurrence of the same batch: %d of %d; final params %s"
      % (WHERE, int(pre_on), bad, steps - 4, h(rec[-1][1])))
