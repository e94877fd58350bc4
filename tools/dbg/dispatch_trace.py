"""Attribute every aten op of one C2 step (flat mode) to the innermost frame inside pointcloudmatters_amd/ using a
TorchDispatchMode (works for forward code and for the Python backward of custom Functions; C++ autograd nodes show up
as '<engine>')."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from pointcloudmatters_amd.bc import BCTrainer, WORKLOADS, build_act_policy, clone_batch, make_act_batch

VIEWS = ("view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "t.default",
         "unbind", "split", "as_strided", "unflatten", "_unsafe_view", "empty", "chunk", "narrow", "size", "stride", "is_", "numel",
         "unsafe_split", "zeros_like", "lift_fresh", "_local_scalar", "sym_", "new_empty", "set_", "record_stream", "resize_")
agg = collections.Counter()

class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            where = "<engine>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "pointcloudmatters_amd/" in fr.filename and "site-packages" not in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("pointcloudmatters_amd/")[-1], fr.lineno, fr.name)
                    break
            shp = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shp = "%s %s" % (tuple(a.shape), str(a.dtype).replace("torch.", ""))
                    break
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                    shp = "[%d x %s]" % (len(a), tuple(a[0].shape))
                    break
            agg[(name.replace("aten.", ""), where, shp)] += 1
        return func(*args, **(kwargs or {}))

dev = torch.device("cuda:0")
WL = os.environ.get("PCM_TRACE_WORKLOAD", "C2")
wl = WORKLOADS[WL]
torch.manual_seed(1000)
if wl.get("policy") == "dp" or WL in ("C3", "C5"):
    from pointcloudmatters_amd.bc import build_dp_policy, make_dp_batch
    from pointcloudmatters_amd.bc.configs import DP_OPTIM
    pol = build_dp_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="flat", optim=dict(DP_OPTIM))
    batches = [make_dp_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
else:
    pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision="bf16", device=dev, mode="flat", optim=dict(accumulate_grad_batches=1))
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, device=dev) for i in range(4)]
for i in range(3):
    tr.training_step(clone_batch(batches[i % 4]))
torch.cuda.synchronize()
b = clone_batch(batches[0])
with Tracer():
    tr.training_step(b)
want = sys.argv[1:] or ["copy_", "_to_copy", "cat", "sum", "add", "fill_", "mul", "div", "sub", "stack", "zero_", "clone", "contiguous"]
tot = 0
for (name, where, shp), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if any(name.startswith(w + ".") or name == w for w in want):
        print("%3d  %-22s %-60s %s" % (n, name, where, shp))
        tot += n
print("total listed", tot, " all ops", sum(agg.values()))
