import torch, time, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev="cuda"
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
B,H,hd=8,8,64
for S in (515,1027,2051):
    q=torch.randn(B,S,H*hd,device=dev).bfloat16().requires_grad_(True); k=torch.randn_like(q).requires_grad_(True); v=torch.randn_like(q).requires_grad_(True)
    def heads(t): return t.view(B,S,H,hd).transpose(1,2)
    for name,be in (("flash",SDPBackend.FLASH_ATTENTION),("efficient",SDPBackend.EFFICIENT_ATTENTION),("math",SDPBackend.MATH)):
        for p in (0.0,0.1):
            try:
                with sdpa_kernel(be):
                    fwd=bench(lambda: F.scaled_dot_product_attention(heads(q),heads(k),heads(v),dropout_p=p))
                    o=F.scaled_dot_product_attention(heads(q),heads(k),heads(v),dropout_p=p); g=torch.randn_like(o)
                    bwd=bench(lambda: torch.autograd.grad(o,(q,k,v),g,retain_graph=True))
                fl=4*S*S*hd*B*H
                print(f"S={S} {name:9s} p={p}: fwd {fwd:7.1f} us ({fl/fwd/1e6:.0f} TF/s)  bwd {bwd:7.1f} us ({2.5*fl/bwd/1e6:.0f} TF/s)")
            except Exception as e:
                print(f"S={S} {name} p={p}: {type(e).__name__} {str(e)[:80]}")
