"""Do the sampling kernels give the same indices when OTHER work runs beside them?  FPS / kNN on a side stream for fixed inputs, 200
times, while the main stream runs GEMMs / elementwise kernels / a captured training graph; every result is compared with the one
computed on an idle device."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd import pointops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, N, M, K = 8, 1024, 512, 16
p = torch.rand(B * N, 3, device=dev)
o = torch.arange(1, B + 1, device=dev, dtype=torch.int32) * N
n_o = torch.arange(1, B + 1, device=dev, dtype=torch.int32) * M
o._pcm_host = [N * (i + 1) for i in range(B)]
n_o._pcm_host = [M * (i + 1) for i in range(B)]
ref_idx = pointops.farthest_point_sampling(p, o, n_o)
ref_knn, _ = pointops.knn_query(K, p, o, p[ref_idx.long()], n_o)
torch.cuda.synchronize()
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
big = torch.randn(64 << 20, device=dev)
for load in ("idle", "gemm", "elementwise", "small kernels"):
    bad_f = bad_k = 0
    for it in range(200):
        if load == "gemm":
            for _ in range(4):
                a @ a
        elif load == "elementwise":
            for _ in range(4):
                big.mul_(1.0001)
        elif load == "small kernels":
            for _ in range(60):
                a[:64].add_(1)
        with torch.cuda.stream(side):
            idx = pointops.farthest_point_sampling(p, o, n_o)
            knn, _ = pointops.knn_query(K, p, o, p[ref_idx.long()], n_o)
            bad_f += int(not torch.equal(idx, ref_idx))
            bad_k += int(not torch.equal(knn, ref_knn))
    torch.cuda.synchronize()
    print(f"{load:14s}: FPS results that differ from the idle run: {bad_f} / 200   kNN: {bad_k} / 200", flush=True)
