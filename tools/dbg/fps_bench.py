import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import pointcloudmatters_amd.pointops as po
from pointcloudmatters_amd.bc import make_act_batch
from oracle import pointops_cpu
dev = torch.device("cuda:0")
for (b, n, m, ragged) in [(8, 1024, 512, False), (8, 2048, 1024, False), (8, 4096, 2048, False), (8, 4096, 2048, True), (8, 700, 350, True)]:
    batch = make_act_batch(b, n, seed=4242, ragged=ragged, device=dev)
    cb = make_act_batch(b, n, seed=4242, ragged=ragged)
    coord, off = batch["pcds"]["coord"], batch["pcds"]["offset"]
    noff = torch.tensor([m * (i + 1) for i in range(b)], dtype=torch.int32, device=dev)
    noff._pcm_host = [m * (i + 1) for i in range(b)]
    got = po.farthest_point_sampling(coord, off, noff)
    want = pointops_cpu.farthest_point_sampling(cb["pcds"]["coord"], cb["pcds"]["offset"], noff.cpu())
    ok = torch.equal(got.cpu(), want)
    ms = bench.timed_events(lambda: po.farthest_point_sampling(coord, off, noff), 10)
    print("T=%s b=%d n=%d m=%d ragged=%s: %.1f us  %.1f ns/pick  %.0f clocks/pick  exact=%s" % (os.environ.get("PCM_FPS_T", "auto"), b, n, m, ragged, ms * 1e3, ms * 1e6 / (m - 1), ms * 1e3 / (m - 1) * 2400, ok), flush=True)
