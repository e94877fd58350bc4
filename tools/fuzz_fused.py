#!/usr/bin/env python
"""Randomized campaign for the fused transformer / U-Net kernels against framework references (tolerances of the unit tests):
attention (random B, H, L, S, key-padding masks with fully masked tiles, dropout with the mask re-derived from the hash),
GroupNorm+Mish (+FiLM, residual, conv bias) and BatchNorm+ReLU.  Usage: python tools/fuzz_fused.py --seconds 120."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    a = ap.parse_args()
    import tests.test_bn_relu_gpu as tb
    import tests.test_small_attn_gpu as ta
    import tests.test_unet_ops_gpu as tu
    from pointcloudmatters_amd.policy import small_attn

    small_attn.MAX_QUERIES, small_attn.MAX_KEYS = 1 << 20, 1 << 20  # exercise the kernels beyond what the policy routes
    orig_supported = small_attn.supported
    t0, n, seed = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng(7_000 + seed)
        B, H = int(rng.integers(1, 5)), int(rng.choice([1, 2, 4, 8]))
        L, S = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 127, 128, 129, 200, 300])), int(rng.choice([1, 7, 32, 33, 95, 100, 128, 260, 515]))
        p = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
        masked = bool(rng.integers(0, 2)) and p == 0.0
        if S < 8 and p > 0:  # one or two keys: the exact gradient is ~0 and only the bf16 rounding of O is left to compare
            p = 0.0
        try:
            ta.run_case(B, H, L, S, masked, packed=False, p=p)
            if L == S and p == 0.0:
                ta.run_case(B, H, L, S, masked, packed=True)
        except AssertionError as e:
            print("ATTENTION MISMATCH", (B, H, L, S, masked, p), e)
            return 1
        shape = (int(rng.integers(1, 6)), int(rng.choice([2, 4, 8, 16])), int(rng.choice([8, 24, 64, 256, 1024])))
        groups = int(rng.choice([g for g in (1, 2, 4, 8) if shape[2] % g == 0]))
        if shape[1] * shape[2] // groups > 6656:  # beyond the kernel's LDS budget: the wrapper falls back to framework ops
            groups = 8
        while groups > 1 and (shape[1] * shape[2] // groups < 16 or shape[2] // groups < 2):  # also: a 1-channel group cancels its conv bias exactly  # 2-element groups normalise to +-1: dx ~ 0, nothing to compare
            groups //= 2
        try:
            tu.test_gn_mish_cl_matches_torch(shape, groups, int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
        except AssertionError as e:
            print("GN_MISH MISMATCH", shape, groups, e)
            return 1
        nrows, c = int(rng.choice([2, 9, 100, 1023, 4097, 20000])), int(rng.choice([4, 24, 64, 128, 512]))
        try:
            tb.test_bn_relu_matches_torch_fp32(nrows, c)
        except AssertionError as e:
            print("BN_RELU MISMATCH", nrows, c, e)
            return 1
        n += 1
        seed += 1
    small_attn.supported = orig_supported
    print(f"fuzz ok: {n} random rounds of attention / gn_mish / bn_relu within the unit-test tolerances")
    return 0


if __name__ == "__main__":
    sys.exit(main())
