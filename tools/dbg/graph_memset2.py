import ctypes, torch, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
def run(n_ints, d32):
    buf = torch.full((n_ints,), 7, device=dev, dtype=torch.int32)
    out = torch.zeros(n_ints, device=dev, dtype=torch.int32)
    def body():
        st = torch.cuda.current_stream().cuda_stream
        rc = (hip.hipMemsetD32Async(buf.data_ptr(), 0, n_ints, st) if d32 else hip.hipMemsetAsync(buf.data_ptr(), 0, n_ints * 4, st))
        assert rc == 0, rc
        buf.add_(1)
        out.copy_(buf)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    for it in range(4):
        g.replay(); torch.cuda.synchronize()
        print("n", n_ints, "d32" if d32 else "d8", "replay", it, out[:8].tolist(), "...", out[-2:].tolist(), "uniq", out.unique().tolist()[:6], flush=True)
for n in (1, 4, 16, 1024):
    run(n, False); run(n, True)
