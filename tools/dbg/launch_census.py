"""Launch census of ONE optimizer step on the HOST wave64 model (tests/wavesim) -- no GPU needed.

What it counts, for the workload's real shapes (default C2: B = 8 x 1024 points -> 512 tokens, bf16 recipe, fused SA layer, flat mode):
  * every launch of a hand-written kernel (the model's `hipLaunchKernelGGL` stand-in notes the kernel expression of each call site);
  * every framework operator that becomes at least one device launch on the GPU (TorchDispatchMode, views / metadata ops dropped),
    split into matrix products (-> hipBLASLt), copies (copy_ / clone / contiguous: the `__amd_rocclr_copyBuffer` / elementwise-copy
    nodes of a replayed step) and the remaining elementwise / reduction operators, each attributed to the innermost product frame.
The step's kernel SEQUENCE is the same in flat and graph mode (a graph is captured from this very code), so the totals are the launch
count a rocprofv3 trace of a replayed step shows, up to the framework operators that take more than one launch.

    python tools/dbg/launch_census.py [--workload C2] [--chain off|short|long] [--copies]

--chain: the MFMA projection chain of csrc/proj_ln.hip (PCM_PROJ_MFMA / PCM_LINEAR_MFMA [/ PCM_PROJ_MFMA_LONG]).
--copies: list every copy-like operator with its call site and shape (the input for removing them)."""
import argparse
import collections
import ctypes
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--chain", default="off", choices=("off", "short", "long"))
ap.add_argument("--copies", action="store_true")
ap.add_argument("--sites", action="store_true", help="list every non-GEMM framework operator with its call site and shape")
ap.add_argument("--tiny", action="store_true", help="B = 2 x 256 points -> 64 tokens (seconds instead of minutes; the encoder then takes attn_small)")
ap.add_argument("--json", default=None)
args = ap.parse_args()
os.environ["PCM_PROJ_MFMA"] = os.environ["PCM_LINEAR_MFMA"] = "0" if args.chain == "off" else "1"
os.environ["PCM_PROJ_MFMA_LONG"] = "1" if args.chain == "long" else "0"

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402
from tests.wavesim.backend import library, simulated_device  # noqa: E402

VIEWS = ("view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "t.default",
         "unbind", "split", "as_strided", "unflatten", "_unsafe_view", "empty", "chunk", "narrow", "size", "stride", "is_", "numel",
         "unsafe_split", "lift_fresh", "_local_scalar", "sym_", "new_empty", "set_", "record_stream", "resize_", "_reshape_alias",
         "result_type", "item", "_has_compatible", "is_same_size")
GEMM = ("mm.", "bmm.", "addmm.", "baddbmm.", "linear.", "matmul.", "_scaled_mm", "addmv", "mv.")
COPY = ("copy_.", "clone.", "contiguous.", "_to_copy.")
ops = collections.Counter()
sites = collections.Counter()
fsites = collections.Counter()
gsites = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "pointcloudmatters_amd/" in fr.filename and "site-packages" not in fr.filename:
            return "%s:%d %s" % (fr.filename.split("pointcloudmatters_amd/")[-1], fr.lineno, fr.name)
    return "<autograd engine>"


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        name = str(func).replace("aten.", "")
        if not any(v in name for v in VIEWS):
            kind = "gemm" if any(name.startswith(g) for g in GEMM) else ("copy" if any(name.startswith(c) for c in COPY) else "elementwise / reduce")
            t = next((x for x in a if isinstance(x, torch.Tensor)), None)
            if kind == "copy" and name.startswith("_to_copy") and t is not None and kw and kw.get("dtype") not in (None, t.dtype):
                kind = "elementwise / reduce"  # a cast kernel, not a memcpy
            if t is not None and t.numel() <= 1 and kind != "gemm":
                kind = kind + " (one element)"
            ops[(kind, name)] += 1
            if kind == "gemm":
                shp2 = " x ".join(str(tuple(x.shape)) for x in a if isinstance(x, torch.Tensor))
                gsites[(name, where(), shp2)] += 1
            if kind != "gemm":
                fsites[(name, where(), tuple(t.shape) if t is not None else (), str(t.dtype).replace("torch.", "") if t is not None else "")] += 1
            if kind.startswith("copy"):
                sites[(name, where(), tuple(t.shape) if t is not None else (), str(t.dtype).replace("torch.", "") if t is not None else "")] += 1
        return func(*a, **(kw or {}))


wl = WORKLOADS[args.workload]
if args.tiny:
    wl = dict(wl, batch=2, n_points=256, pcd_npoints=64)
assert wl["policy"] == "act", "census covers the ACT workloads"
lib = library()
lib.wavesim_census.restype = ctypes.c_long
with simulated_device(claim_cuda=True) as dev:
    torch.manual_seed(0)
    pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
    tr = BCTrainer(pol, total_steps=100, precision=("bf16" if wl["dtype"] == "bf16" else "fp32"), device=dev, mode="flat",
                   optim=dict(accumulate_grad_batches=1))
    batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + i, device=dev) for i in range(3)]
    tr.training_step(clone_batch(batches[0]), prefetch=batches[1])  # caches, lazily built buffers
    lib.wavesim_census_reset()
    with Census():
        tr.training_step(clone_batch(batches[1]), prefetch=batches[2])
    n = lib.wavesim_census(None, 0)
    buf = ctypes.create_string_buffer(n)
    lib.wavesim_census(buf, n)

hand = collections.Counter()
for line in buf.value.decode().splitlines():
    k, c = line.rsplit("\t", 1)
    hand[k.strip("() ")] += int(c)
by_kind = collections.Counter()
for (kind, name), c in ops.items():
    by_kind[kind] += c
print("workload %s, projection chain %s, one optimizer step in flat mode on the host model" % (args.workload, args.chain))
print("hand-written kernel launches: %d in %d kernels" % (sum(hand.values()), len(hand)))
for k, c in sorted(hand.items(), key=lambda kv: -kv[1]):
    print("   %4d  %s" % (c, k))
print("framework operators that launch: %d" % sum(by_kind.values()))
for kind, c in sorted(by_kind.items(), key=lambda kv: -kv[1]):
    print("   %4d  %s" % (c, kind))
    for (k2, name), c2 in sorted(ops.items(), key=lambda kv: -kv[1]):
        if k2 == kind:
            print("           %4d  %s" % (c2, name))
total = sum(hand.values()) + sum(by_kind.values())
print("TOTAL launches (lower bound: one per framework operator): %d" % total)
if args.copies:
    print("copy-like operators by site:")
    for (name, w, shp, dt), c in sorted(sites.items(), key=lambda kv: -kv[1]):
        print("   %3d  %-18s %-58s %s %s" % (c, name, w, shp, dt))
if args.sites:
    print("non-GEMM framework operators by site:")
    for (name, w, shp, dt), c in sorted(fsites.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("   %3d  %-28s %-58s %s %s" % (c, name, w, shp, dt))
if args.sites:
    print("matrix products by site:")
    for (name, w, shp), c in sorted(gsites.items(), key=lambda kv: (kv[0][1], kv[0][2])):
        print("   %3d  %-14s %-52s %s" % (c, name, w, shp))
if args.json:
    import json

    json.dump({"workload": args.workload, "chain": args.chain, "handwritten": dict(hand), "framework": {"%s|%s" % k: v for k, v in ops.items()},
               "total": total}, open(args.json, "w"), indent=1)
