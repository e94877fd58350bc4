"""Ball query: single-kernel paths (pcm_ball_query_b_hip) vs the split path (collect + replay, pcm_ball_query_ws_hip), bit-exact
against each other, timed by HIP events, at the shapes of DESIGN.md section 4 (radius 0.1, nsample 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import timed_events  # noqa: E402
from pointcloudmatters_amd import _lib  # noqa: E402
from pointcloudmatters_amd.bc import make_act_batch  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for name, b, n, mper, ragged in (("C3", 128, 1024, 512, False), ("C5", 32, 4096, 2048, False), ("REF", 8, 4096, 2048, True), ("C2", 8, 1024, 512, False)):
    pc = make_act_batch(b, n, seed=5, ragged=ragged, device=dev)["pcds"]
    xyz, off = pc["coord"].contiguous(), pc["offset"].int()
    import pointcloudmatters_amd.pointops as po
    noff = torch.arange(1, b + 1, device=dev, dtype=torch.int32) * mper
    noff._pcm_host = [mper * (i + 1) for i in range(b)]
    fidx = po.farthest_point_sampling(xyz, pc["offset"], noff)
    q = xyz[fidx.long()].contiguous()
    m, ns = q.shape[0], 16
    i1, d1 = torch.empty(m, ns, dtype=torch.int32, device=dev), torch.empty(m, ns, device=dev)
    i2, d2 = torch.empty_like(i1), torch.empty_like(d1)
    nbytes = int(L.pcm_ball_query_ws_bytes(m))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)

    def old():
        assert L.pcm_ball_query_b_hip(b, m, ns, 0.0, 0.1, xyz.data_ptr(), q.data_ptr(), off.data_ptr(), noff.data_ptr(), i1.data_ptr(), d1.data_ptr(), st) == 0

    def new():
        assert L.pcm_ball_query_ws_hip(b, m, ns, 0.0, 0.1, xyz.data_ptr(), q.data_ptr(), off.data_ptr(), noff.data_ptr(), i2.data_ptr(), d2.data_ptr(),
                                       ws.data_ptr(), nbytes, st) == 0

    old(), new()
    torch.cuda.synchronize()
    same = torch.equal(i1, i2) and torch.equal(d1, d2)
    t_old, t_new = timed_events(old, 20) * 1e3, timed_events(new, 20) * 1e3
    evals = float(m) * (xyz.shape[0] / b)
    print(f"{name}: m={m} pts/cloud~{xyz.shape[0] // b} ws={nbytes / 1e6:.0f} MB  old {t_old:7.1f} us  split {t_new:7.1f} us ({evals / t_new / 1e6:.2f} T evals/s)  identical={same}", flush=True)
