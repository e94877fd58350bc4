import sys, os, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import small_attn
dev="cuda"
B,H=8,8; E=512
L,S=int(sys.argv[1]),int(sys.argv[2])
q=torch.randn(B,L,E,device=dev).bfloat16().requires_grad_(True); k=torch.randn(B,S,E,device=dev).bfloat16().requires_grad_(True); v=torch.randn(B,S,E,device=dev).bfloat16().requires_grad_(True)
go=torch.randn(B,L,E,device=dev).bfloat16()
for _ in range(30):
    out=small_attn.small_attention(q,k,v,None,H,0.0)
    torch.autograd.grad(out,(q,k,v),go)
    o2=F.scaled_dot_product_attention(q.view(B,L,H,64).transpose(1,2),k.view(B,S,H,64).transpose(1,2),v.view(B,S,H,64).transpose(1,2))
    torch.autograd.grad(o2,(q,k,v),go.view(B,L,H,64).transpose(1,2))
torch.cuda.synchronize()
