"""Device time of the own transformer-tail kernels under graph replay (no profiler inflation), at the C2 row counts."""
import sys, torch
sys.path.insert(0, "/root/repo")
import torch.nn as nn
from pointcloudmatters_amd.policy import fused_ops
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
E, F = 512, 32
l1, l2, norm = nn.Linear(E, F).to(dev), nn.Linear(F, E).to(dev), nn.LayerNorm(E).to(dev)
dh, do = nn.Dropout(0.1), nn.Dropout(0.1)
ctx = fused_ops.FusedContext(dev); ctx.set_step(1)
for R in (800, 816, 4120):
    x = torch.randn(R // 8, 8, E, device=dev)
    y = torch.randn(R // 8, 8, E, device=dev).bfloat16()
    with fused_ops.activate(ctx):
        t_ffn_f = timed(lambda: fused_ops.ffn_ln(x, l1, l2, norm, dh, do))
        t_ffn_fb = 0.0
        t_drln_f = timed(lambda: fused_ops.drln(x, y, norm, do))
        t_drln_fb = 0.0
    print("R=%d  ffn fwd %.1f us, fwd+bwd %.1f us | drln fwd %.1f us, fwd+bwd %.1f us" % (R, t_ffn_f, t_ffn_fb, t_drln_f, t_drln_fb))
