"""kNN kernels at the bench shapes: device time per call (torch.profiler) and evaluations per second.
PCM_KNN_TWOPASS=0/1 python tools/mb/mb_knn.py
(the switches exist only in the microbenchmark build: `make -C pointcloudmatters_amd/csrc mb`, then
PCM_POINTOPS_LIB=$PWD/pointcloudmatters_amd/lib_mb/libpcm_pointops.so)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pointcloudmatters_amd.pointops as po  # noqa: E402

dev = "cuda"
shapes = {"C2": (8, 1024, 512), "C4": (8, 2048, 1024), "REF": (8, 4096, 2048), "C5": (32, 4096, 1024), "N16K": (4, 16384, 4096)}
for name, (b, n, mq) in shapes.items():
    g = torch.Generator(device=dev).manual_seed(1)
    xyz = torch.rand(b * n, 3, device=dev, generator=g)
    off = torch.arange(1, b + 1, device=dev, dtype=torch.int32) * n
    noff = torch.arange(1, b + 1, device=dev, dtype=torch.int32) * mq
    fidx = po.farthest_point_sampling(xyz, off, noff)
    q = xyz[fidx.long()]
    for _ in range(3):
        idx, d2 = po.knn_query(16, xyz, off, q, noff)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            po.knn_query(16, xyz, off, q, noff)
        torch.cuda.synchronize()
    rows = {}
    for e in prof.key_averages():
        if "pcm_knn" in e.key:
            k = e.key.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            rows[k] = e.self_device_time_total / e.count
    tot = sum(rows.values())
    evals = b * mq * n
    print(f"{name:5s} b={b} n={n} m={b*mq}: " + "  ".join(f"{k} {v:.1f}us" for k, v in rows.items()) + f"  -> {evals / tot / 1e6:.3f} T evals/s", flush=True)
