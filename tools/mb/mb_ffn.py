"""fp32 (csrc/ffn.hip) vs matrix-core (csrc/ffn_mfma.hip) feed-forward sub-layer, forward and backward, HIP-event timed, at the
row counts of the ACT step (decoder / CVAE encoder 800-816 rows, encoder 4120 at C2, 8216 at C4, 16408 at REF)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import timed_events  # noqa: E402
from pointcloudmatters_amd import _lib  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
f32 = dict(dtype=torch.float32, device=dev)
E, F = 512, 32
st = torch.cuda.current_stream().cuda_stream
for R in (800, 816, 4120, 8216, 16408):
    x = torch.randn(R, E, **f32)
    w1, b1, w2, b2 = torch.randn(F, E, **f32) * 0.05, torch.zeros(F, **f32), torch.randn(E, F, **f32) * 0.05, torch.zeros(E, **f32)
    g, bt = torch.ones(E, **f32), torch.zeros(E, **f32)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    hd, s, out = torch.empty(R, F, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32)
    mean, rstd = torch.empty(R, **f32), torch.empty(R, **f32)
    dout, dx, dy, dh = torch.randn(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, F, **f32)
    part = torch.empty(max(L.pcm_ffn_ln_blocks(R), L.pcm_ffn_ln_mfma_blocks(R)) * (3 * E + F), **f32)
    res = {}
    for name, fwd, bwd in (("fp32", L.pcm_ffn_ln_forward2_hip, L.pcm_ffn_ln_backward2_hip),
                           ("mfma", L.pcm_ffn_ln_mfma_forward_hip, L.pcm_ffn_ln_mfma_backward_hip)):
        def f(st=st):
            assert fwd(R, E, F, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g.data_ptr(), bt.data_ptr(), 1e-5,
                       0.1, 0.1, seed.data_ptr(), 1, 2, hd.data_ptr(), s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0, 0, 0, 0,
                       st) == 0

        def b(st=st):
            assert bwd(R, E, F, dout.data_ptr(), 0, x.data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), hd.data_ptr(), w1.data_ptr(),
                       w2.data_ptr(), g.data_ptr(), 0.1, 0.1, seed.data_ptr(), 2, dx.data_ptr(), dy.data_ptr(), dh.data_ptr(), part.data_ptr(), 0,
                       st) == 0

        f()
        b()

        def graphed(fn, n=20):
            """n launches captured into one hipGraph and replayed: device time per launch without the host's launch cost (a ctypes
            call with 26 arguments takes ~9 us, more than these kernels)."""
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                st_ = torch.cuda.current_stream().cuda_stream
                with torch.cuda.graph(g, stream=side):
                    for _ in range(n):
                        fn(st_)
            return timed_events(g.replay, 20) * 1e3 / n

        res[name] = (graphed(lambda s_: f(s_)), graphed(lambda s_: b(s_)))
    fb, bb = R * (E * 12 + F * 4), R * (E * 20 + F * 8)
    print(f"R={R:6d}  fwd fp32 {res['fp32'][0]:6.1f} us  mfma {res['mfma'][0]:6.1f} us ({fb / res['mfma'][0] / 1e3:5.0f} GB/s)   "
          f"bwd fp32 {res['fp32'][1]:6.1f} us  mfma {res['mfma'][1]:6.1f} us ({bb / res['mfma'][1] / 1e3:5.0f} GB/s)", flush=True)
