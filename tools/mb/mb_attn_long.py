import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import small_attn, fused_ops
small_attn.MAX_QUERIES = 1 << 20; small_attn.MAX_KEYS = 1 << 20
dev="cuda"; B,H=8,8; E=512
S=int(sys.argv[1]); p=float(sys.argv[2])
q=torch.randn(B,S,E,device=dev).bfloat16().requires_grad_(True); k=torch.randn_like(q).requires_grad_(True); v=torch.randn_like(q).requires_grad_(True)
go=torch.randn(B,S,E,device=dev).bfloat16()
ctx=fused_ops.FusedContext(dev)
with fused_ops.activate(ctx):
    for _ in range(10):
        out=small_attn.small_attention(q,k,v,None,H,p)
        if len(sys.argv) > 3: torch.autograd.grad(out,(q,k,v),go)
        o2=F.scaled_dot_product_attention(q.view(B,S,H,64).transpose(1,2),k.view(B,S,H,64).transpose(1,2),v.view(B,S,H,64).transpose(1,2),dropout_p=p)
        if len(sys.argv) > 3: torch.autograd.grad(o2,(q,k,v),go.view(B,S,H,64).transpose(1,2))
torch.cuda.synchronize()
ref=F.scaled_dot_product_attention(q.view(B,S,H,64).transpose(1,2).float(),k.view(B,S,H,64).transpose(1,2).float(),v.view(B,S,H,64).transpose(1,2).float()).transpose(1,2).reshape(B,S,E)
if p == 0: print("max err vs fp32:", (small_attn.small_attention(q,k,v,None,H,0.0).float()-ref).abs().max().item())
