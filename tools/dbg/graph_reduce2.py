import torch, sys
dev = torch.device("cuda:0")
torch.manual_seed(0)

def capture(body):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        res = body()
    return g, res

# 1. one column-sum per graph, replay-only loop (no eager work in between), first failing replay
for nred in (1, 2, 4):
    x = torch.randn(1024, 512, device=dev).to(torch.bfloat16)
    def body():
        return [ (x * float(i + 1)).sum(dim=0) for i in range(nred) ]
    g, res = capture(body)
    first = None
    nbad = 0
    for it in range(300):
        x.copy_(torch.randn(1024, 512, device=dev).to(torch.bfloat16))
        g.replay()
        torch.cuda.synchronize()
        want = x.float().sum(dim=0)
        ok = all(torch.allclose(r.float(), want * (i + 1), rtol=3e-2, atol=0.5) for i, r in enumerate(res))
        if not ok:
            nbad += 1
            if first is None: first = it
    print("nred", nred, "first bad replay", first, "nbad", nbad, flush=True)

# 2. memset node: zeros via cudaMemsetAsync inside a graph
buf = torch.empty(1024, device=dev, dtype=torch.int32)
def body2():
    buf.zero_()
    buf.add_(1)
    return buf
g, _ = capture(body2)
first = None
for it in range(300):
    g.replay(); torch.cuda.synchronize()
    if not bool((buf == 1).all()) and first is None:
        first = it
print("memset+add graph: first bad", first, "final unique", buf.unique().tolist()[:5], flush=True)

# 3. same as 1 but re-running the eager sum between replays only (is the eager kernel's own state involved?)
x = torch.randn(1024, 512, device=dev).to(torch.bfloat16)
def body3():
    return x.sum(dim=0)
g, res = capture(body3)
first = None
for it in range(300):
    x.copy_(torch.randn(1024, 512, device=dev).to(torch.bfloat16))
    e = x.sum(dim=0)   # eager twin
    g.replay(); torch.cuda.synchronize()
    if not torch.allclose(res.float(), x.float().sum(dim=0), rtol=3e-2, atol=0.5) and first is None:
        first = it
    if not torch.allclose(e.float(), x.float().sum(dim=0), rtol=3e-2, atol=0.5):
        print("EAGER bf16 sum wrong at", it); break
print("with eager twin: first bad", first, flush=True)
# 4. fp32 sum over (4120, 512) etc: which shapes take the global-reduce path and fail?
for shape in [(4120, 512), (8192, 512), (2048, 512), (1024, 96), (1024, 2048), (515, 512), (800, 512), (1024, 7)]:
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(*shape, device=dev).to(dt)
        def body4():
            return x.sum(dim=0)
        g, res = capture(body4)
        first = None
        for it in range(150):
            x.copy_(torch.randn(*shape, device=dev).to(dt))
            g.replay(); torch.cuda.synchronize()
            if not torch.allclose(res.float(), x.float().sum(dim=0), rtol=3e-2, atol=1.0) and first is None:
                first = it
        print(shape, dt, "first bad", first, flush=True)
