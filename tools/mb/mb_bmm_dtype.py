import torch
a=torch.randn(8,512,515,device="cuda").bfloat16(); b=torch.randn(8,515,256,device="cuda").bfloat16()
try:
    o=torch.bmm(a,b,out_dtype=torch.float32); print("bmm out_dtype ok", o.dtype, (o-torch.bmm(a.float(),b.float())).abs().max().item())
except Exception as e: print("bmm out_dtype failed:", type(e).__name__, str(e)[:300])
try:
    o=torch.mm(a[0],b[0],out_dtype=torch.float32); print("mm out_dtype ok", o.dtype)
except Exception as e: print("mm out_dtype failed:", type(e).__name__, str(e)[:300])
