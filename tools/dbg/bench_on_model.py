"""Development aid: run bench.py's per-kernel tables (kernel_rooflines -- ~60 direct C-ABI launches with hand-built arguments) on the
HOST wave64 model (tests/wavesim) at a tiny shape.  Times are meaningless (every launch "takes" 1 ms); the point is that every entry
point is called with valid arguments and every byte / flop formula evaluates -- a check of bench.py's table code without a GPU.
    python tools/dbg/bench_on_model.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from tests.wavesim.backend import simulated_device

calls = []


def timed(fn, iters, warmup=3):
    fn()
    calls.append(1)
    return 1.0


bench.timed_events = timed
wl = dict(policy="act", batch=2, n_points=256, pcd_npoints=64, ragged=False, dtype="bf16")
for tok_bf16 in (False, True):
    bench.TOKENIZER_BF16 = tok_bf16
    with simulated_device(claim_cuda=True) as dev:
        rows = bench.kernel_rooflines(wl, dev)
    print("tokenizer", "bf16" if tok_bf16 else "fp32", ":", len(rows), "table rows,", len(calls), "timed launches")
    for k in sorted(rows):
        if "sa_fwd" in k or "bn_relu" in k or "sa_bwd2" in k:
            print("   ", k, {kk: rows[k][kk] for kk in list(rows[k])[:3]})

# ---- the whole of bench.main() on a tiny workload: argument parsing, policy / trainer construction (graph mode falls back to flat on
# the model: no hipGraphs), the timed loop, the JSON line with its `config` / `roofline` fields
if "--main" in sys.argv:
    from pointcloudmatters_amd.bc import WORKLOADS

    WORKLOADS["TINY"] = dict(WORKLOADS["C2"], batch=2, n_points=256, pcd_npoints=64)
    bench.TOKENIZER_BF16 = False
    sys.argv = ["bench.py", "--workload", "TINY", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-hbm-tables",
                "--tables-out", "/tmp/bench_on_model_tables.json"] + [a for a in sys.argv[1:] if a not in ("--main",)]
    with simulated_device(claim_cuda=True):
        torch.cuda.is_available = lambda: True
        torch.cuda.set_device = lambda d: None
        torch.cuda.empty_cache = lambda: None

        class _Torch:  # bench.py's view of torch: "cuda" devices are the host
            def __getattr__(self, name):
                return getattr(torch, name)

            @staticmethod
            def device(kind, index=None):
                return torch.device("cpu") if kind == "cuda" else torch.device(kind)

        bench.torch = _Torch()
        bench.main()
