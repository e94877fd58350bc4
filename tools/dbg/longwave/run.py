import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liblongwave.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "longwave.hip"), "-o", SO])
L = ctypes.CDLL(SO)
L.longwave_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.rand(8 * 128 * 8, device=dev)
amat = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
(amat @ amat); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    with torch.cuda.graph(g, stream=s2):
        for _ in range(400):
            amat @ amat
torch.cuda.synchronize()
side = torch.cuda.Stream()
for mode in (0, 1, 2):
    ref = torch.empty(8 * 128, device=dev)
    L.longwave_launch(8, 512, mode, x.data_ptr(), ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for load in ("idle", "gemm graph"):
        outs = []
        for it in range(200):
            if load != "idle":
                g.replay()
            with torch.cuda.stream(side):
                o = torch.empty(8 * 128, device=dev)
                L.longwave_launch(8, 512, mode, x.data_ptr(), o.data_ptr(), side.cuda_stream)
            outs.append(o)
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs)
        print(f"mode {mode} beside {load}: {bad} / 200 results differ from the idle run", flush=True)
