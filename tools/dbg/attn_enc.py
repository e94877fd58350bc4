import sys, torch, functools
print = functools.partial(print, flush=True)
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from pointcloudmatters_amd.policy import small_attn
small_attn.MAX_QUERIES = 1024
dev = torch.device("cuda:0")
def timed(fn, n=10):
    """sum of device kernel time per call (torch.profiler)"""
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return sum(e.device_time_total for e in prof.key_averages()) / n
B, H, L, E = 8, 8, 515, 512
for L in (515, 514, 512, 258, 1027):
    torch.manual_seed(0)
    q, k, v = [torch.randn(B, L, E, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(3)]
    go = torch.randn(B, L, E, device=dev, dtype=torch.bfloat16)
    def sh(t): return t.view(B, L, H, 64).transpose(1, 2)
    def flash():
        o = F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(B, L, E)
        return o
    def small():
        return small_attn.small_attention(q, k, v, None, H, 0.0)
    print('start', L, flush=True)
    ref = F.scaled_dot_product_attention(sh(q.float()), sh(k.float()), sh(v.float())).transpose(1, 2).reshape(B, L, E)
    gref = torch.autograd.grad(ref, (q, k, v), go.float())
    print('ref ok', flush=True)
    o1 = small(); torch.cuda.synchronize(); print('small fwd ok', flush=True); g1 = torch.autograd.grad(o1, (q, k, v), go)
    torch.cuda.synchronize(); print('small bwd ok', flush=True)
    o0 = o1; g0 = g1
    err = lambda a, b: float((a.float() - b.float()).abs().max())
    print(L, "fwd err small %.4f flash %.4f | dq %.4f/%.4f dk %.4f/%.4f dv %.4f/%.4f" % (err(o1, ref), err(o0, ref), err(g1[0], gref[0]), err(g0[0], gref[0]),
          err(g1[1], gref[1]), err(g0[1], gref[1]), err(g1[2], gref[2]), err(g0[2], gref[2])))
    tf0, tf1 = 0.0, timed(lambda: small())
    def fb(f):
        o = f(); torch.autograd.grad(o, (q, k, v), go)
    tb0, tb1 = 0.0, timed(lambda: fb(small))
    print("   fwd flash %.1f small %.1f | fwd+bwd flash %.1f small %.1f us" % (tf0, tf1, tb0, tb1))
