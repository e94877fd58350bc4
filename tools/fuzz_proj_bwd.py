#!/usr/bin/env python
"""Randomized campaign for round 6's backward kernels of csrc/proj_ln.hip on the HOST wave64 model (no GPU): random row counts / widths /
dropout rates / optional second gradient for pcm_proj_drln_mfma_backward (dx, dy bit-equal to pcm_drln_backward2, da against the fp64 product)
and random (R, N, K, pos_cols, residual, strides) for pcm_linear_mfma_backward.      python tools/fuzz_proj_bwd.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import random

import torch

from tests.wavesim.backend import simulated_device
from tests import test_proj_ln_gpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T.DEV = "cpu"
bad = 0
with simulated_device(claim_cuda=False):
    for c in range(cases):
        if c % 2 == 0:
            R, E, K = rng.choice([1, 2, 15, 16, 17, 31, 100, 257, 800, 1030]), rng.choice([256, 512, 768, 1024]), rng.choice([256, 512, 1024])
            p = rng.choice([0.0, 0.1, 0.5])
            dout, dout2, s, mean, rstd, gamma, W = T._bwd_inputs(R, E, K, rng.randrange(1 << 30))
            d2 = dout2 if rng.random() < 0.5 else None
            seed = torch.tensor([rng.randrange(1 << 40)], dtype=torch.int64) if p > 0 else None
            site = rng.randrange(64)
            dx0, dy0, sums0, _ = T._row_kernel(dout, d2, s, mean, rstd, gamma, p, seed, site)
            dx, dy, sums, _, da = T._chain_bwd(dout, d2, s, mean, rstd, gamma, W, p, seed, site)
            want = dy0.double() @ W.double()
            ok = (torch.equal(dx, dx0) and torch.equal(dy.view(torch.int16), dy0.view(torch.int16))
                  and (sums - sums0).abs().max().item() <= 2e-5 * sums0.abs().max().item() + 1e-6
                  and (da.double() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-6)
            what = ("chain", R, E, K, p, d2 is not None)
        else:
            R, K = rng.choice([1, 3, 16, 33, 100, 800, 1000]), rng.choice([256, 512, 1024])
            N = 32 * rng.randrange(1, 49)
            pos_cols = rng.choice([0, N, N + 5] + [32 * rng.randrange(0, N // 32 + 1)])
            g = torch.Generator().manual_seed(rng.randrange(1 << 30))
            wide = torch.randn(R, N + 8 * rng.randrange(0, 9), generator=g).bfloat16()
            dy = wide[:, :N]
            W = (torch.randn(N, K, generator=g) / N ** 0.5).bfloat16()
            dres = torch.randn(R, K, generator=g) if rng.random() < 0.5 else None
            dx, dpos = T._lin_bwd(dy, W, dres, pos_cols, True)
            want = dy.double() @ W.double()
            wpos = dy[:, :pos_cols].double() @ W[:pos_cols].double() if pos_cols < N else want
            tol = 3e-6 * N ** 0.5 * max(1.0, want.abs().max().item())
            ok = ((dx.double() - (want + (dres.double() if dres is not None else 0))).abs().max().item() <= tol
                  and (dpos.double() - wpos).abs().max().item() <= tol)
            what = ("linear", R, N, K, pos_cols, dres is not None)
        if not ok:
            bad += 1
            print("MISMATCH", what, flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
