"""Where the CPU baseline of bench.py (cpu_baseline: reference op order on the host, C oracle pointops, fp32) spends a C2 step:
forward time per component (module hooks), backward, optimizer.  python tools/dbg/cpu_profile.py [threads]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pointops_cpu  # noqa: E402
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count())
torch.set_num_threads(threads)
os.environ["OMP_NUM_THREADS"] = str(threads)
wl = WORKLOADS["C2"]
torch.manual_seed(0)
policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], pointops=pointops_cpu, sa_impl="reference")
trainer = BCTrainer(policy, total_steps=100, precision="fp32", device="cpu", optim=dict(accumulate_grad_batches=1))
batch = make_act_batch(wl["batch"], wl["n_points"], seed=1000, device="cpu")
acc = {}


def timed(name, mod):
    def pre(m, a):
        m._t0 = time.perf_counter()

    def post(m, a, o):
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - m._t0

    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post)


timed("PointNet forward", policy.backbone)
timed("CVAE encoder forward", policy.encoder)
timed("transformer forward (4 encoder + 7 decoder layers)", policy.transformer)
for name in ("farthest_point_sampling", "knn_query"):
    real = getattr(pointops_cpu, name)

    def wrap(*a, _real=real, _name=name, **k):
        t0 = time.perf_counter()
        out = _real(*a, **k)
        acc["oracle " + _name] = acc.get("oracle " + _name, 0.0) + time.perf_counter() - t0
        return out

    setattr(pointops_cpu, name, wrap)
real_bw = torch.Tensor.backward


def bw(self, *a, **k):
    t0 = time.perf_counter()
    real_bw(self, *a, **k)
    acc["backward (all components)"] = acc.get("backward (all components)", 0.0) + time.perf_counter() - t0


torch.Tensor.backward = bw
real_step = trainer.optimizer.step


def st(*a, **k):
    t0 = time.perf_counter()
    out = real_step(*a, **k)
    acc["clip + AdamW"] = acc.get("clip + AdamW", 0.0) + time.perf_counter() - t0
    return out


trainer.optimizer.step = st
trainer.training_step(clone_batch(batch))
acc.clear()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    trainer.training_step(clone_batch(batch))
total = (time.perf_counter() - t0) / n
print("threads %d  step %.3f s  (%.2f samples/s)" % (threads, total, wl["batch"] / total))
known = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-55s %7.3f s  %5.1f %%" % (k, v / n, 100 * v / n / total))
    known += v / n
print("  %-55s %7.3f s  %5.1f %%" % ("SA layer forward (group + Linear + BN + ReLU + max), loss, rest", total - known, 100 * (total - known) / total))
