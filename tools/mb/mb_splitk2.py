import torch, time
dev="cuda"
def bench(f, n=50):
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize(); t=time.perf_counter()
    g.replay(); torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for rows in (4120, 816, 8192, 131072):
  for (m, k) in [(512,512),(1024,512),(64,64),(512,128)]:
    go=torch.randn(rows,m,device=dev,dtype=torch.bfloat16); x=torch.randn(rows,k,device=dev,dtype=torch.bfloat16)
    w=torch.randn(m,k,device=dev,dtype=torch.bfloat16)
    r={}
    r["plain dW"]=bench(lambda: go.t()@x)
    for s in (4,8,16,32,64):
        if rows % s: continue
        chunk=rows//s
        def f():
            return torch.bmm(go.view(s,chunk,m).transpose(1,2), x.view(s,chunk,k)).sum(0, dtype=torch.float32)
        r["S%d"%s]=bench(f)
    r["fwd"]=bench(lambda: x@w.t()); r["dgrad"]=bench(lambda: go@w)
    print(rows,(m,k), {k_: "%.1f"%v for k_,v in r.items()})
