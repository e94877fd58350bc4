import torch, time
dev="cuda"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for rows in (131072, 32768, 8192):
  for (m, k) in [(64,6),(64,64),(128,64),(512,128),(96,512),(96,96),(128,96)]:
    go=torch.randn(rows,m,device=dev,dtype=torch.bfloat16); x=torch.randn(rows,k,device=dev,dtype=torch.bfloat16)
    r={}
    r["plain"]=bench(lambda: go.t()@x)
    for chunk in (512,1024,2048,4096):
        if rows % chunk: continue
        s=rows//chunk
        def f():
            p=torch.bmm(go.view(s,chunk,m).transpose(1,2), x.view(s,chunk,k))
            return p.sum(0, dtype=torch.float32)
        r["bmm%d"%chunk]=bench(f)
    ref=(go.float().t()@x.float()); a=(go.t()@x).float(); s=rows//1024
    b=torch.bmm(go.view(s,1024,m).transpose(1,2), x.view(s,1024,k)).sum(0,dtype=torch.float32)
    err=lambda v: ((v-ref).abs().max()/ref.abs().max()).item()
    print(rows,(m,k), {k_: "%.0fus"%v for k_,v in r.items()}, "err plain %.1e bmm %.1e"%(err(a),err(b)))
