"""How many queries the fast kNN kernel hands to the exact kernel (ties among the nsample+1 smallest, buffer overflow):
PCM_KNN_SKIP_EXACT=1 python tools/mb/mb_knn_flags.py
(the switches exist only in the microbenchmark build: `make -C pointcloudmatters_amd/csrc mb`, then
PCM_POINTOPS_LIB=$PWD/pointcloudmatters_amd/lib_mb/libpcm_pointops.so)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pointcloudmatters_amd.pointops as po  # noqa: E402
from pointcloudmatters_amd.bc import make_act_batch, make_dp_batch  # noqa: E402

dev = "cuda"
cases = {}
g = torch.Generator(device=dev).manual_seed(1)
for name, sizes, mq in (("uniform 8x4096", [4096] * 8, 2048), ("ragged 2k..8k", [2500, 8000, 4100, 6000, 3000, 7000, 5000, 4500], 2048),
                        ("16k", [16384] * 2, 4096)):
    xyz = torch.rand(sum(sizes), 3, device=dev, generator=g)
    cases[name] = (xyz, sizes, mq)
b = make_dp_batch(16, 4096, seed=3, ragged=True, device=dev)
pc = b["obs"]["pcds"]
o = pc["offset"].tolist()
cases["make_dp_batch ragged"] = (pc["coord"], [o[0]] + [o[i] - o[i - 1] for i in range(1, len(o))], 1024)
b = make_act_batch(8, 4096, seed=3, ragged=True, device=dev)
pc = b["pcds"]
o = pc["offset"].tolist()
cases["make_act_batch ragged"] = (pc["coord"], [o[0]] + [o[i] - o[i - 1] for i in range(1, len(o))], 2048)
for name, (xyz, sizes, mq) in cases.items():
    off = torch.tensor(sizes, device=dev).cumsum(0).int()
    noff = (torch.arange(1, len(sizes) + 1, device=dev) * mq).int()
    fidx = po.farthest_point_sampling(xyz.contiguous(), off, noff)
    q = xyz[fidx.long()].contiguous()
    from pointcloudmatters_amd.pointops.query import knn_query_raw

    idx, d2 = knn_query_raw(16, xyz.contiguous(), off, q, noff)
    torch.cuda.synchronize()
    print(f"{name:24s} queries {q.shape[0]:6d} marked {(d2[:, 0] < 0).sum().item():6d}  sizes {min(sizes)}..{max(sizes)}")
