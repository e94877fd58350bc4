"""FPS device time per call at the bench shapes (torch.profiler).  PCM_FPS_SMALL_T=64|128 python tools/mb/mb_fps.py
(the switches exist only in the microbenchmark build: `make -C pointcloudmatters_amd/csrc mb`, then
PCM_POINTOPS_LIB=$PWD/pointcloudmatters_amd/lib_mb/libpcm_pointops.so)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pointcloudmatters_amd.pointops as po  # noqa: E402

dev = "cuda"
for name, (b, n, mq) in {"C2": (8, 1024, 512), "C3": (128, 1024, 256), "C4": (8, 2048, 1024), "C5": (32, 4096, 2048), "roll": (1, 4096, 2048), "roll2": (2, 4096, 2048)}.items():
    g = torch.Generator(device=dev).manual_seed(1)
    xyz = torch.rand(b * n, 3, device=dev, generator=g)
    off = torch.arange(1, b + 1, device=dev, dtype=torch.int32) * n
    noff = torch.arange(1, b + 1, device=dev, dtype=torch.int32) * mq
    ref = None
    for _ in range(3):
        ref = po.farthest_point_sampling(xyz, off, noff)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            po.farthest_point_sampling(xyz, off, noff)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "pcm_fps" in e.key:
            t = e.self_device_time_total / e.count
            print(f"{name} b={b} n={n} m={mq}: {t:.1f} us = {t / (mq - 1) * 1e3:.0f} ns/pick  checksum {int(ref.long().sum())}  {e.key[:60]}", flush=True)
