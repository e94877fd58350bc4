"""policy/deferred.py counters over a few C2-shaped hybrid steps: python tools/dbg/deferred_stats.py [mode]"""
import sys

import torch

from pointcloudmatters_amd.bc import BCTrainer, build_act_policy, clone_batch, make_act_batch
from pointcloudmatters_amd.policy import deferred

mode = sys.argv[1] if len(sys.argv) > 1 else "hybrid"
torch.manual_seed(0)
pol = build_act_policy(pcd_npoints=512, sa_impl="fused").cuda()
tr = BCTrainer(pol, total_steps=100, precision="bf16", device="cuda", mode=mode)
b = make_act_batch(8, 1024, seed=1, device="cuda")
for i in range(6):
    before = dict(deferred.STATS)
    tr.training_step(clone_batch(b))
    torch.cuda.synchronize()
    print(i, {k: deferred.STATS[k] - before[k] for k in before}, flush=True)
