import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.policy.rows_linear import weight_grad
import pointcloudmatters_amd.policy.rows_linear as RL
dev="cuda"
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for rows,(m,k) in ((4120,(512,512)),(4120,(1024,512)),(8192,(512,128)),(8192,(64,64))):
    go=torch.randn(rows,m,device=dev).bfloat16(); x=torch.randn(rows,k,device=dev).bfloat16()
    res={}
    res["plain"]=bench(lambda: go.t()@x)
    for s in (2,4,8,16):
        if rows % s: continue
        ch=rows//s
        a,b=go.view(s,ch,m).transpose(1,2), x.view(s,ch,k)
        res[f"S{s} f32out"]=bench(lambda: torch.bmm(a,b,out_dtype=torch.float32))
        res[f"S{s} bf16out"]=bench(lambda: torch.bmm(a,b))
    res["weight_grad"]=bench(lambda: weight_grad(go,x,torch.bfloat16))
    print(rows,(m,k),{k_: "%.1f"%v for k_,v in res.items()})
