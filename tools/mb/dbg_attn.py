import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy import small_attn
from tests.test_small_attn_gpu import reference
torch.manual_seed(0)
B,H,L,S=1,1,32,64
E=H*64
q=torch.randn(B,L,E,device="cuda").bfloat16(); k=torch.randn(B,S,E,device="cuda").bfloat16(); v=torch.randn(B,S,E,device="cuda").bfloat16()
for sel in ([63],[0],[5],[36],[32,33,34,35],list(range(32,64)),list(range(0,32))):
    kpm=torch.zeros(B,S,dtype=torch.bool,device="cuda"); kpm[:, sel]=True
    out=small_attn.small_attention(q,k,v,kpm,H,0.0).float(); want=reference(q,k,v,kpm,H); nomask=reference(q,k,v,None,H)
    # which single-key mask would explain the output best?
    best=None
    for j in range(S):
        km=torch.zeros(B,S,dtype=torch.bool,device="cuda"); km[:, j]=True
        e=(out-reference(q,k,v,km,H)).abs().max().item()
        if best is None or e<best[0]: best=(e,j)
    print(sel[:4], "err vs masked ref %.3f, vs unmasked ref %.3f; best single-key explanation: key %d (err %.3f)"%((out-want).abs().max().item(), (out-nomask).abs().max().item(), best[1], best[0]))
