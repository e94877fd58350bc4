"""Event-timed A/B of csrc/attn_flash.hip against the framework's scaled_dot_product_attention (bf16), forward and
backward, B*H = 64 (the encoder self-attention of the BASELINE configs).  python tools/mb/mb_flash.py [p_drop]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import fused_ops, small_attn  # noqa: E402

dev = "cuda"
B, H, E = 8, 8, 512
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1


def timed(fn, iters=20):
    """GPU time per call: the calls are captured into one hipGraph (no host gaps between the launches) and replayed."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    from pointcloudmatters_amd import _graphs

    def body():
        for _ in range(iters):
            fn()

    g, _ = _graphs.captured(body)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / (5 * iters) * 1e3


ctx = fused_ops.FusedContext(dev)
for S in (515, 1027, 2051):
    q, k, v = (torch.randn(B, S, E, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    go = torch.randn(B, S, E, device=dev).bfloat16()
    heads = lambda t: t.view(B, S, H, 64).transpose(1, 2)  # noqa: E731
    with fused_ops.activate(ctx):
        f_flash = timed(lambda: small_attn.small_attention(q, k, v, None, H, p))
        out = small_attn.small_attention(q, k, v, None, H, p)
        b_flash = timed(lambda: torch.autograd.grad(out, (q, k, v), go, retain_graph=True))
    f_sdpa = timed(lambda: F.scaled_dot_product_attention(heads(q), heads(k), heads(v), dropout_p=p))
    o2 = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), dropout_p=p)
    b_sdpa = timed(lambda: torch.autograd.grad(o2, (q, k, v), heads(go), retain_graph=True))
    fl = 4 * B * H * S * S * 64
    print("S=%4d p=%.2f  fwd: flash %7.1f us (%6.1f TF/s)  sdpa %7.1f us | bwd: flash %7.1f us (%6.1f TF/s, 7 GEMM units)  sdpa %7.1f us"
          % (S, p, f_flash, fl / f_flash / 1e6, f_sdpa, b_flash, 3.5 * fl / b_flash / 1e6, b_sdpa), flush=True)
