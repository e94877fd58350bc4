import torch, sys
dev = torch.device("cuda:0")
rows, cols, dt = 1024, 512, torch.bfloat16
x = torch.zeros(rows, cols, device=dev, dtype=dt)
def body():
    a = (x * 2.0)
    t = torch.empty(1 << 20, device=dev); del t
    s1 = a.sum(dim=0)
    b = (a + 1.0)
    s2 = b.sum(dim=0)
    s3 = (b.float().reshape(-1, cols)).sum(dim=0)
    return a, s1, s2, s3
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    res = body()
shown = 0
for it in range(1, 300):
    v = float(it % 64) / 64.0      # exactly representable
    x.fill_(v)
    y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
    z = y.to(torch.bfloat16).sum(dim=0)
    g.replay()
    want = body()
    torch.cuda.synchronize()
    a, s1, s2, s3 = res
    got = (a.float().mean().item() / 2, s1.float().mean().item() / (2 * rows), (s3.float().mean().item() / rows - 1) / 2)
    if any(abs(q - v) > 0.01 for q in got) and shown < 12:
        shown += 1
        print("it", it, "v", v, "graph saw: a->%.4f s1->%.4f s3->%.4f" % got, " a uniform:", bool((a == a.flatten()[0]).all()), "s1 uniform", bool((s1 == s1[0]).all()), flush=True)
print("done, mismatching iterations shown:", shown)
