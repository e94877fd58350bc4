"""attn_small.hip vs attn_flash.hip on the SHORT query sets of the ACT step (decoder / CVAE encoder): kernel durations
from the roctracer trace (host launch gaps excluded).  python tools/mb/mb_attn_short.py [p_drop]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import fused_ops, small_attn  # noqa: E402

dev = "cuda"
B, H, E = 8, 8, 512
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
shapes = [(100, 100), (102, 102), (100, 515), (100, 2051), (515, 515), (2051, 2051)]
if len(sys.argv) > 2:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[2:]]
ctx = fused_ops.FusedContext(dev)
for L, S in shapes:
    q = torch.randn(B, L, E, device=dev).bfloat16().requires_grad_(True)
    k, v = (torch.randn(B, S, E, device=dev).bfloat16().requires_grad_(True) for _ in range(2))
    go = torch.randn(B, L, E, device=dev).bfloat16()
    for name, frm in (("small", 1 << 20), ("flash", 1)):
        small_attn.FLASH_FROM = frm
        rows = {}
        for tag, grad in (("", go), ("(dO=0)", torch.zeros_like(go))):
            with fused_ops.activate(ctx):
                for _ in range(2):
                    out = small_attn.small_attention(q, k, v, None, H, p)
                    torch.autograd.grad(out, (q, k, v), grad)
                torch.cuda.synchronize()
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(10):
                        out = small_attn.small_attention(q, k, v, None, H, p)
                        torch.autograd.grad(out, (q, k, v), grad)
                    torch.cuda.synchronize()
            for e in prof.key_averages():
                if "pcm_attn" in e.key:
                    short = e.key.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("pcm_attn_", "").replace("_kernel", "")
                    rows[short + tag] = e.self_device_time_total / e.count
        print("L=%4d S=%4d p=%.2f %s: " % (L, S, p, name) + "  ".join("%s %.1f" % kv for kv in sorted(rows.items())), flush=True)
