"""csrc/gemm_small.hip against the library GEMM (F.linear, bf16) at the projection shapes of the ACT step: device time per product
inside a replayed hipGraph (20 dependent products per graph), and the largest difference of the results."""
import os
import sys

import torch
import torch.nn.functional as F

import ctypes
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from bench import timed_events  # noqa: E402

SO = os.path.join(HERE, "libgemm_small.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "gemm_small.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(HERE, "gemm_small.hip"), "-o", SO])
L = ctypes.CDLL(SO)
_l, _p, _i = ctypes.c_long, ctypes.c_void_p, ctypes.c_int
L.pcm_gemm_bf16_nt_hip.argtypes = [_l, _l, _l, _p, _l, _p, _l, _p, _i, _p, _l, _i, _p]
dev = torch.device("cuda", 0)
bf = dict(dtype=torch.bfloat16, device=dev)


def graphed(fn, n=20):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(n):
                fn(st)
    return timed_events(g.replay, 20) * 1e3 / n


shapes = [(800, 512, 512), (816, 512, 512), (800, 1536, 512), (1632, 1536, 512), (800, 1024, 512), (8, 512, 512), (100, 512, 512),
          (4120, 512, 512), (4120, 1536, 512), (4120, 3584, 512), (800, 512, 1024), (800, 512, 256), (16408, 512, 512)]
for M, N, K in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, **bf)
    w = torch.randn(N, K, **bf) * 0.05
    b = torch.randn(N, **bf)
    c = torch.empty(M, N, **bf)
    ref = torch.empty(M, N, **bf)

    def ours(st):
        rc = L.pcm_gemm_bf16_nt_hip(M, N, K, a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), 0, c.data_ptr(), N, 0, st)
        assert rc == 0, rc

    def lib(st):
        torch.addmm(b, a, w.t(), out=ref)

    t_ours, t_lib = graphed(ours), graphed(lib)
    exact = (a.float() @ w.float().t() + b.float())
    e_ours = (c.float() - exact).abs().max().item()
    e_lib = (ref.float() - exact).abs().max().item()
    print(f"M={M:6d} N={N:5d} K={K:5d}  ours {t_ours:6.2f} us  library {t_lib:6.2f} us   max|err| vs fp32: ours {e_ours:.4f} library {e_lib:.4f}"
          f"   {2e-6 * M * N * K / t_ours:7.1f} TFLOP/s", flush=True)
