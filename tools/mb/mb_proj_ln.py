"""csrc/proj_ln.hip against what it replaces, by graph replay (device time per launch, no host launch cost):
  A  out_proj + residual + norm:  library bf16 product (F.linear) + pcm_drln_forward_hip      vs  pcm_proj_drln_mfma_forward_hip
  B  packed in-projection:        pcm_add_cast2_hip + the doubled-row product [x + pos ; x] W^T  vs  pcm_linear_mfma_forward_hip (fp32 x + pos in)
  C  query projection:            pcm_add_cast2_hip + library product                          vs  pcm_linear_mfma_forward_hip
  D  backward of A (800 / 816 rows): pcm_drln_backward2_hip + library product dy @ W           vs  pcm_proj_drln_mfma_backward_hip (round 6)
  E  backward of B's input (800 / 816): batched product 3 x (R, E) @ (E, E) + pcm_add4_cast2_hip  vs  pcm_linear_mfma_backward_hip (round 6)
at the row counts of the ACT step (decoder 800, CVAE encoder 816, encoder 4120 at C2).  Written in round 5 while the GPU pool was closed:
the FIRST thing to run when it opens --   python tools/mb/mb_proj_ln.py > gpurun_out/mb_proj_ln.log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import timed_events  # noqa: E402
from pointcloudmatters_amd import _lib  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
f32, bf = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
E = 512


def graphed(fn, n=20):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn(torch.cuda.current_stream().cuda_stream)
    return timed_events(g.replay, 20) * 1e3 / n


for R in (800, 816, 4120):
    a = torch.randn(R, E, **f32).bfloat16()
    W, b = (torch.randn(E, E, **f32) / E ** 0.5).bfloat16(), torch.zeros(E, **bf)
    W3, b3 = (torch.randn(3 * E, E, **f32) / E ** 0.5).bfloat16(), torch.zeros(3 * E, **bf)
    x, pos = torch.randn(R, E, **f32), torch.randn(100 if R % 100 == 0 else R, E, **f32)
    gamma, beta = torch.ones(E, **f32), torch.zeros(E, **f32)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    s, out, mean, rstd = torch.empty(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, **f32), torch.empty(R, **f32)
    y16 = torch.empty(R, E, **bf)
    pair = torch.empty(2, R, E, **bf)
    y3, y1 = torch.empty(R, 3 * E, **bf), torch.empty(R, E, **bf)

    def lib_a(st):
        torch.nn.functional.linear(a, W, b, )  # result allocated by the framework, like in the step
        assert L.pcm_drln_forward_hip(R, E, 1, x.data_ptr(), y16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, seed.data_ptr(), 3,
                                      s.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), st) == 0

    def new_a(st):
        assert L.pcm_proj_drln_mfma_forward_hip(R, E, E, a.data_ptr(), E, W.data_ptr(), b.data_ptr(), 1, x.data_ptr(), gamma.data_ptr(),
                                                beta.data_ptr(), 1e-5, 0.1, seed.data_ptr(), 3, s.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), 0, 0, 0, 0, st) == 0

    def lib_b(st):
        assert L.pcm_add_cast2_hip(x.numel(), pos.numel(), x.data_ptr(), pos.data_ptr(), pair[0].data_ptr(), pair[1].data_ptr(), st) == 0
        torch.nn.functional.linear(pair.view(2 * R, E), W3, b3)

    def new_b(st):
        assert L.pcm_linear_mfma_forward_hip(R, 3 * E, E, x.data_ptr(), 1, E, 0, pos.data_ptr(), pos.numel(), 2 * E, W3.data_ptr(), b3.data_ptr(), 1,
                                             y3.data_ptr(), 1, 3 * E, pair[0].data_ptr(), pair[1].data_ptr(), st) == 0

    def lib_c(st):
        assert L.pcm_add_cast2_hip(x.numel(), pos.numel(), x.data_ptr(), pos.data_ptr(), pair[0].data_ptr(), 0, st) == 0
        torch.nn.functional.linear(pair[0], W, b)

    def new_c(st):
        assert L.pcm_linear_mfma_forward_hip(R, E, E, x.data_ptr(), 1, E, 0, pos.data_ptr(), pos.numel(), E, W.data_ptr(), b.data_ptr(), 1,
                                             y1.data_ptr(), 1, E, pair[0].data_ptr(), 0, st) == 0

    dout, dx, dy16, da16 = torch.randn(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **bf), torch.empty(R, E, **bf)
    torch.nn.functional.layer_norm(x, (E,))  # any s / mean / rstd will do for timing
    mean.copy_(x.mean(1)), rstd.copy_((x.var(1, unbiased=False) + 1e-5).rsqrt())
    part_old = torch.empty(L.pcm_drln_blocks(R) * 3 * E, **f32)
    part_new = torch.empty(max(1, L.pcm_proj_drln_mfma_backward_blocks(R)) * 3 * E, **f32)
    sums, db16 = torch.empty(3, E, **f32), torch.empty(E, **bf)

    def lib_d(st):
        assert L.pcm_drln_backward2_hip(R, E, 1, dout.data_ptr(), 0, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), 0.1,
                                        seed.data_ptr(), 3, dx.data_ptr(), dy16.data_ptr(), part_old.data_ptr(), sums.data_ptr(), db16.data_ptr(), st) == 0
        dy16 @ W

    def new_d(st):
        assert L.pcm_proj_drln_mfma_backward_hip(R, E, E, dout.data_ptr(), 0, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), 0.1,
                                                 seed.data_ptr(), 3, W.data_ptr(), dx.data_ptr(), dy16.data_ptr(), da16.data_ptr(), E,
                                                 part_new.data_ptr(), sums.data_ptr(), db16.data_ptr(), st) == 0

    d3y = torch.randn(R, 3 * E, **f32).bfloat16()
    dres, dx32, dpos32 = torch.randn(R, E, **f32), torch.empty(R, E, **f32), torch.empty(R, E, **f32)

    def lib_e(st):
        d3 = torch.bmm(torch.as_strided(d3y, (3, R, E), (E, 3 * E, 1)), W3.view(3, E, E))
        assert L.pcm_add4_cast2_hip(dx32.numel(), d3[0].data_ptr(), d3[1].data_ptr(), d3[2].data_ptr(), dres.data_ptr(), dx32.data_ptr(), dpos32.data_ptr(), st) == 0

    def new_e(st):
        assert L.pcm_linear_mfma_backward_hip(R, 3 * E, E, d3y.data_ptr(), 3 * E, W3.data_ptr(), dres.data_ptr(), dx32.data_ptr(), dpos32.data_ptr(), 2 * E, st) == 0

    ALL = (lib_a, new_a, lib_b, new_b, lib_c, new_c, lib_d, new_d) + ((lib_e, new_e) if R <= 1024 else ())
    for fn in ALL:
        fn(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = {fn.__name__: graphed(fn) for fn in ALL}
    print(f"R={R:5d}  A out_proj+res+norm: library pair {t['lib_a']:6.1f} us  proj_ln {t['new_a']:6.1f} us   "
          f"B in-projection: add_cast + doubled product {t['lib_b']:6.1f} us  linear_mfma {t['new_b']:6.1f} us   "
          f"C query projection: {t['lib_c']:6.1f} us  linear_mfma {t['new_c']:6.1f} us   "
          f"D backward of A: drln_bwd + reduce + dy @ W {t['lib_d']:6.1f} us  chain kernel + reduce {t['new_d']:6.1f} us"
          + (f"   E in-projection input gradient: bmm + add4 {t['lib_e']:6.1f} us  linear_mfma_backward {t['new_e']:6.1f} us" if 'lib_e' in t else ""), flush=True)
