"""FPS / kNN on a side stream, 300 times without host synchronisation, while the main stream replays the captured C2 training graph;
results (device-side clones) are compared with the idle-device result at the end.  PCM_FPS_SMALL_T selects the FPS variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd import pointops  # noqa: E402
from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
torch.manual_seed(1000)
policy = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(policy, total_steps=2000, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + 97 * i, ragged=False, device=dev) for i in range(2)]
for i in range(3):
    tr.training_step(clone_batch(batches[0]))
torch.cuda.synchronize()
pc = batches[1]["pcds"]
p, o = pc["coord"], pc["offset"]
n_o = policy._new_offsets(o)
ref = pointops.farthest_point_sampling(p, o, n_o)
ref_knn, _ = pointops.knn_query(16, p, o, p[ref.long()], n_o)
torch.cuda.synchronize()
side = torch.cuda.Stream()
load = os.environ.get("LOAD", "graph")
small = torch.zeros(1 << 16, device=dev)
amat = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
g_tiny = g_gemm = None
if load in ("tiny_graph", "gemm_graph"):
    g = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            for _ in range(400):
                if load == "tiny_graph":
                    small.add_(1.0)
                else:
                    amat @ amat
    torch.cuda.synchronize()
    g_tiny = g
out = []
for it in range(300):
    if load == "graph":
        tr._graph[0].replay()
    elif load == "step":
        tr.training_step(clone_batch(batches[0]))
    elif load in ("tiny_graph", "gemm_graph"):
        g_tiny.replay()
    elif load == "tiny_eager":
        for _ in range(400):
            small.add_(1.0)
    side.wait_stream(torch.cuda.current_stream()) if os.environ.get("WAIT") == "1" else None
    with torch.cuda.stream(side):
        idx = pointops.farthest_point_sampling(p, o, n_o)
        knn, _ = pointops.knn_query(16, p, o, p[ref.long()], n_o)
    out.append((idx, knn))
torch.cuda.synchronize()
bf = sum(int(not torch.equal(a, ref)) for a, _ in out)
bk = sum(int(not torch.equal(b, ref_knn)) for _, b in out)
M = wl["pcd_npoints"]
shown = 0
for a, _ in out:
    if not torch.equal(a, ref) and shown < 6:
        A, R = a.view(-1, M), ref.view(-1, M)
        print("   differing clouds (cloud, first differing pick, count):", [(c, int((A[c] != R[c]).nonzero()[0]), int((A[c] != R[c]).sum())) for c in range(A.shape[0]) if not torch.equal(A[c], R[c])])
        shown += 1
print(f"load={load} PCM_FPS_SMALL_T={os.environ.get('PCM_FPS_SMALL_T', '128')}: FPS differs in {bf} / 300, kNN in {bk} / 300")
