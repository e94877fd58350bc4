"""Where a tile of csrc/ffn_mfma.hip's forward spends its time: the kernel built with -DPCM_FFN_CLOCKS stamps clock64() per wave of
workgroup 0 at its phase boundaries.  Prints, per wave, the clocks between stamps (shader clocks) for R = 800 and 4120 rows."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
SO = os.path.join(HERE, "libffn_clocks.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DPCM_FFN_CLOCKS",
                       os.path.join(CSRC, "ffn_mfma.hip"), os.path.join(CSRC, "ffn.hip"), "-o", SO])
L = ctypes.CDLL(SO)
P, I, F_, LG, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long, ctypes.c_uint
L.pcm_ffn_ln_mfma_forward_hip.argtypes = [LG, I, I, P, P, P, P, P, P, P, F_, F_, F_, P, U, U, P, P, P, P, P, P, LG, P, P, P]
L.pcm_ffn_clocks_read.argtypes = [P]
dev = torch.device("cuda", 0)
f32 = dict(dtype=torch.float32, device=dev)
E, F = 512, 32
names = ["issue loads + dropout keys (waits for the seed)", "mfma1 (waits for x, W1)", "cross-wave sum", "relu/dropout a", "mfma2 + residual + dropout b", "epilogue loads",
         "row statistics (2 barriers)", "normalise + stores"]
for R in (800, 4120):
    x = torch.randn(R, E, **f32)
    w1, b1, w2, b2 = torch.randn(F, E, **f32) * 0.05, torch.zeros(F, **f32), torch.randn(E, F, **f32) * 0.05, torch.zeros(E, **f32)
    g, bt = torch.ones(E, **f32), torch.zeros(E, **f32)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    hd, s, out = torch.empty(R, F, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32)
    mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        rc = L.pcm_ffn_ln_mfma_forward_hip(R, E, F, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g.data_ptr(), bt.data_ptr(),
                                           1e-5, 0.1, 0.1, seed.data_ptr(), 1, 2, hd.data_ptr(), s.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), None, 0, None, None, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (2 * 8 * 16))()
    assert L.pcm_ffn_clocks_read(buf) == 0
    print(f"R = {R}: clocks between stamps, per wave of workgroup 0 (forward)")
    for it in range(2):
        print("  pass", it, "(cold code)" if it == 0 else "(same code again, another tile: cold data)")
        for w in range(4):
            c = [buf[(it * 8 + w) * 16 + i] for i in range(8)]
            print("    wave", w, " ".join(f"{c[i + 1] - c[i]:7d}" for i in range(7)), " total", c[7] - c[0])
    print("   phases:", " | ".join(names[:7]))
