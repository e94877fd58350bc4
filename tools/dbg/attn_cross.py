import sys, torch, functools
print = functools.partial(print, flush=True)
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from pointcloudmatters_amd.policy import small_attn
small_attn.MAX_KEYS = 8192
dev = torch.device("cuda:0")
def timed(fn, n=10):
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return sum(e.device_time_total for e in prof.key_averages()) / n
B, H, E = 8, 8, 512
for L, S in ((100, 2051), (100, 1027), (100, 515), (100, 4099)):
    torch.manual_seed(0)
    q = torch.randn(B, L, E, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    k, v = [torch.randn(B, S, E, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(2)]
    go = torch.randn(B, L, E, device=dev, dtype=torch.bfloat16)
    def sh(t): return t.view(B, t.shape[1], H, 64).transpose(1, 2)
    def flash(): return F.scaled_dot_product_attention(sh(q), sh(k), sh(v), dropout_p=0.0).transpose(1, 2).reshape(B, L, E)
    def small(): return small_attn.small_attention(q, k, v, None, H, 0.0)
    ref = F.scaled_dot_product_attention(sh(q.float()), sh(k.float()), sh(v.float())).transpose(1, 2).reshape(B, L, E)
    gref = torch.autograd.grad(ref, (q, k, v), go.float())
    o1 = small(); g1 = torch.autograd.grad(o1, (q, k, v), go)
    err = lambda a, b: float((a.detach().float() - b.detach().float()).abs().max())
    print(L, S, "err fwd %.4f dq %.4f dk %.4f dv %.4f" % (err(o1, ref), err(g1[0], gref[0]), err(g1[1], gref[1]), err(g1[2], gref[2])))
    def fb(f):
        o = f(); o.backward(go); q.grad = k.grad = v.grad = None
    print("   fwd+bwd: flash %.1f us  small %.1f us ; fwd: flash %.1f small %.1f" % (timed(lambda: fb(flash)), timed(lambda: fb(small)), timed(flash), timed(small)))
