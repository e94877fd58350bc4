import torch, sys
dev = torch.device("cuda:0")

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

def run(churn, eager_mm, eager_body, eager_sum, rows=1024, cols=512, iters=200):
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    def body():
        a = (x * 2.0)
        if churn:
            t = torch.empty(1 << 20, device=dev); del t
        s1 = a.sum(dim=0)
        b = (a + 1.0)
        s2 = b.sum(dim=0)
        return s1, s2
    g, res = capture(body)
    first, nbad = None, 0
    for it in range(iters):
        x.copy_(torch.randn(rows, cols, device=dev).to(torch.bfloat16))
        if eager_mm:
            y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
        if eager_sum:
            z = torch.randn(1024, 512, device=dev).to(torch.bfloat16).sum(dim=0)
        g.replay()
        if eager_body:
            want = body()
        torch.cuda.synchronize()
        w1 = (x.double() * 2).sum(0); w2 = (x.double() * 2 + 1).sum(0)
        e = max((res[0].double() - w1).abs().max().item(), (res[1].double() - w2).abs().max().item())
        if e > 8.0:
            nbad += 1
            if first is None: first = (it, e)
    print(f"churn={churn} mm={eager_mm} body={eager_body} sum={eager_sum}: first bad {first}, nbad {nbad}", flush=True)

run(1, 1, 1, 1)
run(0, 1, 1, 1)
run(1, 0, 1, 1)
run(1, 1, 0, 1)
run(1, 1, 1, 0)
run(0, 0, 0, 0)
run(0, 0, 1, 0)
run(0, 0, 0, 1)
run(0, 0, 0, 0, iters=2000)
