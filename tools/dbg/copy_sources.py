"""Which Python lines issue the device-to-device copies / clones / casts of a C2 training step (flat mode = the launches that
graph mode replays)?  python tools/dbg/copy_sources.py [op-substring ...]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda:0")
wl = WORKLOADS["C2"]
torch.manual_seed(0)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
batch = make_act_batch(wl["batch"], wl["n_points"], seed=1000, device=dev)
for _ in range(3):
    tr.training_step(clone_batch(batch))
torch.cuda.synchronize()
import traceback

from torch.utils._python_dispatch import TorchDispatchMode

want = sys.argv[1:] or ["copy_", "clone", "_to_copy", "add", "add_", "sum", "cat", "fill_", "zero_", "mul", "stack", "index_select", "slice_backward", "select_backward"]
agg = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in want:
            big = any(torch.is_tensor(a) and a.is_cuda for a in args)
            if big:
                frames = [f for f in traceback.extract_stack() if "pointcloudmatters_amd" in f.filename and "_python_dispatch" not in f.filename]
                where = "%s:%d %s" % (frames[-1].filename.split("pointcloudmatters_amd/")[-1], frames[-1].lineno, frames[-1].name) if frames else "(autograd engine / torch internals)"
                agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.training_step(clone_batch(batch))
    torch.cuda.synchronize()
for (name, where), n in agg.most_common(70):
    print("%3d  %-16s %s" % (n, name, where))
print("total", sum(agg.values()))
