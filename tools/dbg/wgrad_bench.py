import sys, torch
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.policy import rows_linear as RL
dev = torch.device("cuda:0")
def timed(fn, n=20):
    """device time per call: n calls captured in one hipGraph, replayed"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
for rows, m, k in [(4120, 512, 512), (4120, 1024, 512), (4120, 32, 512), (4120, 512, 32), (4120, 3584, 512), (8192, 512, 512), (8192, 64, 64), (8192, 128, 64), (8192,512,128), (4112, 512, 512)]:
    go = torch.randn(rows, m, device=dev, dtype=torch.bfloat16)
    x = torch.randn(rows, k, device=dev, dtype=torch.bfloat16)
    out = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
    t_plain = timed(lambda: torch.mm(go.t(), x, out=out))
    res = {}
    for s in (2, 4, 5, 8, 10, 16):
        if rows % s: continue
        chunk = rows // s
        a, b = go.view(s, chunk, m).transpose(1, 2), x.view(s, chunk, k)
        part = torch.empty(s, m, k, device=dev)
        def f():
            p = torch.bmm(a, b, out_dtype=torch.float32)
            return p.sum(0)
        res[s] = timed(f)
    cur = timed(lambda: RL.weight_grad(go, x, torch.bfloat16, out=out))
    print(rows, m, k, "plain %.1f  current %.1f  split " % (t_plain, cur), {s: round(v, 1) for s, v in res.items()})
