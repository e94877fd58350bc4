import torch, time
dev="cuda"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for (cout, ck, bt) in [(2048,10240,256),(2048,5120,256),(2048,20480,256),(1024,20480,512),(2048,10240,1024)]:
    go=torch.randn(bt,cout,device=dev,dtype=torch.bfloat16); cols=torch.randn(bt,ck,device=dev,dtype=torch.bfloat16)
    w=torch.randn(cout,ck,device=dev,dtype=torch.bfloat16)
    fl=2*cout*ck*bt
    r={}
    r["go.t()@cols"]=bench(lambda: go.t()@cols)
    r["(cols.t()@go).t()"]=bench(lambda: (cols.t()@go))
    got=go.t().contiguous(); colst=cols.t().contiguous()
    r["got_c@cols"]=bench(lambda: got@cols)
    r["got_c@colst_c.t()"]=bench(lambda: got@colst.t())
    r["fwd cols@w.t()"]=bench(lambda: cols@w.t())
    r["dgrad go@w"]=bench(lambda: go@w)
    r["cols.t()@go (no .t)"]=bench(lambda: cols.t()@go)
    r["colst_c@go"]=bench(lambda: colst@go)
    r["f32 out"]=bench(lambda: torch.mm(go.t().float(), cols.float())) if bt<=256 else 0
    gp=torch.nn.functional.pad(go,(0,0,0,256)); cp=torch.nn.functional.pad(cols,(0,0,0,256))
    r["padK+256"]=bench(lambda: gp.t()@cp)
    print((cout,ck,bt), {k: "%.0fus %.0fTF"%(v, fl/v/1e6) for k,v in r.items()})
