import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
for name, wl, c, h in [("C2", dict(batch=8, n_points=1024, pcd_npoints=512, ragged=False), 512, 512),
                       ("C3", dict(batch=128, n_points=1024, pcd_npoints=512, ragged=False), 96, 96),
                       ("C5", dict(batch=32, n_points=4096, pcd_npoints=2048, ragged=False), 96, 96),
                       ("REF", dict(batch=8, n_points=4096, pcd_npoints=2048, ragged=True), 512, 512)]:
    kr = bench.kernel_rooflines(wl, dev, c_feat=c, hidden=h)
    print(name)
    for k, v in kr.items():
        if "sa_" in k or "fps" in k or "knn" in k or "group" in k:
            print("   %-40s %8.1f us  %8.1f MB  %7.0f GB/s" % (k, v["ms"] * 1e3, v["algorithmic_bytes"] / 1e6, v["achieved_GBs"]))
