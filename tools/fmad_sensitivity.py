"""How much of "bit-exact FPS / kNN indices" depends on the arithmetic contract (squared distances un-contracted)?

The reference's kernels write  d = dx*dx + dy*dy + dz*dz  (sampling_cuda_kernel.cu:54, knn_query_cuda_kernel.cu:91,
ball_query_cuda_kernel.cu:91); a CUDA compiler's default -fmad=true would fuse that into two FMAs, this repository (oracle AND library,
-ffp-contract=off) rounds every product and sum on its own.  No CUDA device is here to ask, so this tool measures the SENSITIVITY on the
CPU: the oracle built both ways (oracle/libpcm_oracle.so vs the -DPCM_ORACLE_FMAD variant libpcm_oracle_fmad.so) on the bench
workloads' synthetic clouds, counting the FPS picks, clouds, and kNN rows that change.  CPU only; test infrastructure, never the product.
    python tools/fmad_sensitivity.py            # prints one JSON line per workload shape
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lib as olib  # noqa: E402
from oracle import pointops_cpu as po  # noqa: E402
from tests.util import make_clouds, new_offsets  # noqa: E402


def _load(name):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), name], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", name))
    for fn, args in olib._SIGS.items():
        getattr(lib, fn).argtypes = args
        getattr(lib, fn).restype = ctypes.c_int
    return lib


def run(lib, xyz, off, noff, k):
    olib._LIB = lib
    sel = po.farthest_point_sampling(xyz, off, noff)
    return sel


def main():
    plain, fused = _load("libpcm_oracle.so"), _load("libpcm_oracle_fmad.so")
    shapes = [("C2", 8, 1024, 512), ("C4", 8, 2048, 1024), ("REF", 8, 4096, 2048)]
    for name, b, n, m in shapes:
        picks = changed_picks = changed_clouds = first_div = 0
        rows = rows_list = rows_set = rows_dist = 0
        for seed in range(4):
            xyz, off = make_clouds([n] * b, seed=1000 + seed)
            noff = new_offsets([m] * b)
            a, f = run(plain, xyz, off, noff, 16), run(fused, xyz, off, noff, 16)
            a2, f2 = a.view(b, m), f.view(b, m)
            picks += a.numel()
            changed_picks += int((a2 != f2).sum())
            d = (a2 != f2).any(1)
            changed_clouds += int(d.sum())
            # the SAME queries for both (the plain picks): isolates the neighbour search from the sampling
            q = xyz[a.long()].contiguous()
            olib._LIB = plain
            i1, d1 = po.knn_query_raw(16, xyz, off, q, noff)
            olib._LIB = fused
            i2, d2 = po.knn_query_raw(16, xyz, off, q, noff)
            rows += i1.shape[0]
            rows_list += int((i1 != i2).any(1).sum())
            rows_set += int((i1.sort(1).values != i2.sort(1).values).any(1).sum())
            rows_dist += int((d1 != d2).any(1).sum())
        print(json.dumps({"shape": name, "clouds": 4 * b, "points": n, "picks_per_cloud": m,
                          "fps_clouds_whose_pick_sequence_changes": changed_clouds, "fps_picks_changed": changed_picks, "fps_picks": picks,
                          "knn_rows": rows, "knn_rows_with_another_index_list": rows_list, "knn_rows_with_another_index_SET": rows_set,
                          "knn_rows_with_another_dist2_value": rows_dist}), flush=True)
    olib._LIB = None


if __name__ == "__main__":
    main()
