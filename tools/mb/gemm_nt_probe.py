import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
lib = ctypes.CDLL("/root/repo/tools/mb/libgemm_nt_probe.so")
lib.gemm_nt_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
for M, N, K in [(800, 512, 512), (800, 1024, 512), (816, 512, 512), (4120, 512, 512), (4120, 1024, 512), (4120, 3584, 512), (8192, 512, 128), (8192, 128, 64)]:
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    ref = torch.nn.functional.linear(A, W, b)
    t_ref = timed(lambda: torch.nn.functional.linear(A, W, b))
    res = {}
    for v in range(6):
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def f():
            rc = lib.gemm_nt_launch(v, M, N, K, A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        f(); torch.cuda.synchronize()
        err = (C.float() - ref.float()).abs().max().item()
        res[v] = (round(timed(f), 2), round(err, 4))
    print(M, N, K, "hipblaslt %.2f us | probe" % t_ref, res)
