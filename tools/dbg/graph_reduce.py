"""Does a torch column-sum (global-reduce path: staging buffer + semaphores + captured memset) replay correctly in a hipGraph?"""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for rows, cols, dt in [(1024, 512, torch.bfloat16), (1024, 1024, torch.bfloat16), (512, 512, torch.bfloat16), (4096, 512, torch.bfloat16), (1024, 512, torch.float32)]:
    x = torch.randn(rows, cols, device=dev).to(dt)
    outs = []
    def body():
        a = (x * 2.0)
        t = torch.empty(1 << 20, device=dev)  # churn the pool
        del t
        s1 = a.sum(dim=0)
        b = (a + 1.0)
        s2 = b.sum(dim=0)
        s3 = (b.float().reshape(-1, cols)).sum(dim=0)
        return s1, s2, s3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        res = body()
    bad = 0
    for it in range(200):
        x.copy_(torch.randn(rows, cols, device=dev).to(dt))
        # some eager work between replays
        y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
        z = y.to(torch.bfloat16).sum(dim=0)
        g.replay()
        want = body()
        torch.cuda.synchronize()
        xd = x.double()
        truth = [(xd * 2).sum(0), (xd * 2 + 1).sum(0), (xd * 2 + 1).sum(0)]
        for j, (r, w, t) in enumerate(zip(res, want, truth)):
            eg = (r.double() - t).abs().max().item()
            ee = (w.double() - t).abs().max().item()
            if eg > 8 or ee > 8:
                bad += 1
                if bad <= 6:
                    print("   it", it, "out", j, "graph err", eg, "eager err", ee)
    print(rows, cols, dt, "bad replays:", bad, flush=True)
