"""hipGraph memset nodes under eager activity between replays: do they stop working?"""
import ctypes, torch, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")

def run(eager, n_ints, iters=400):
    buf = torch.full((n_ints,), 7, device=dev, dtype=torch.int32)
    out = torch.zeros(n_ints, device=dev, dtype=torch.int32)
    def body():
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n_ints * 4, st)
        assert rc == 0, rc
        buf.add_(1)          # -> 1 if the memset ran, grows otherwise
        out.copy_(buf)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    first = None
    for it in range(iters):
        if "mm" in eager:
            y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
        if "sum" in eager:
            z = torch.randn(2048, 2048, device=dev).to(torch.bfloat16).sum(dim=0)
        if "memset" in eager:
            t = torch.empty(64, device=dev, dtype=torch.int32)
            hip.hipMemsetAsync(t.data_ptr(), 0, 256, torch.cuda.current_stream().cuda_stream)
        if "alloc" in eager:
            ts = [torch.empty(1 << 22, device=dev) for _ in range(4)]
        g.replay()
        torch.cuda.synchronize()
        if not bool((out == 1).all()) and first is None:
            first = (it, out.unique().tolist()[:4])
    print("eager=%-22s n_ints=%d first bad: %s" % (eager, n_ints, first), flush=True)

for n in (16, 1024):
    run("", n)
    run("mm", n)
    run("sum", n)
    run("memset", n)
    run("mm,sum", n)
    run("mm,sum,alloc", n)
