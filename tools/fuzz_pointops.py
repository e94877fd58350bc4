#!/usr/bin/env python
"""Long randomized parity campaign of the index kernels against the CPU oracle (bit-exact), beyond the 24 seeded cases of
tests/test_pointops_fuzz_gpu.py: FPS, kNN, ball query and random ball query on ragged batches with tie-heavy data.
Usage (on the GPU box): python tools/fuzz_pointops.py --seconds 240 [--seed0 0].  Exit status 1 on the first mismatch.
--model runs the same campaign on the HOST wave64 model (tests/wavesim: the kernel sources compiled for the CPU; no GPU needed, sizes capped
so that a layout takes seconds): evidence about the kernels' logic, also under WAVESIM_ORDER=reverse|shuffle:N (another lane interleaving)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.util import make_clouds, new_offsets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--model", action="store_true", help="run on the host wave64 model instead of cuda:0")
    a = ap.parse_args()
    if a.model:
        from tests.wavesim.backend import simulated_device

        with simulated_device() as dev:
            return campaign(a, dev, scales=[2, 9, 70, 400, 1100, 2100], m_cap=600)
    return campaign(a, torch.device("cuda", 0), scales=[2, 9, 70, 400, 1100, 2100, 4200, 9000], m_cap=2500)


def campaign(a, d, scales, m_cap):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref
    from pointcloudmatters_amd.pointops.query import ball_query_raw, knn_query_raw, random_ball_query_raw

    t0, n_cases, seed = time.time(), 0, a.seed0
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng(50_000 + seed)
        b = int(rng.integers(1, 10))
        scale = int(rng.choice(scales))
        sizes = [int(rng.integers(1, scale + 1)) for _ in range(b)]
        ms = [int(rng.integers(1, min(max(2 * s, 2), m_cap) + 1)) for s in sizes]
        mode = str(rng.choice(["uniform", "lattice", "dup"]))
        lattice = float(rng.choice([0.003, 0.02, 0.1, 0.3]))
        xyz, off = make_clouds(sizes, seed=seed, mode=mode, lattice=lattice)
        noff = new_offsets(ms)
        tag = (seed, sizes, ms, mode, lattice)
        want = ref.farthest_point_sampling(xyz, off, noff)
        got = po.farthest_point_sampling(xyz.to(d), off.to(d), noff.to(d))
        if not torch.equal(got.cpu(), want):
            print("FPS MISMATCH", tag)
            return 1
        new_xyz = xyz[want.long()].contiguous()
        ns = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 64, 128]))
        wi, wd = ref.knn_query_raw(ns, xyz, off, new_xyz, noff)
        gi, gd = knn_query_raw(ns, xyz.to(d), off.to(d), new_xyz.to(d), noff.to(d))
        if not (torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)):
            print("KNN MISMATCH", tag, ns)
            return 1
        if max(sizes) <= 2048:  # the reference's candidate stack holds 2048 entries
            nsb = int(rng.choice([1, 8, 16, 32]))
            rmax, rmin = float(rng.choice([0.05, 0.2, 0.6])), float(rng.choice([0.0, 0.02]))
            wi, wd = ref.ball_query_raw(nsb, rmax, rmin, xyz, off, new_xyz, noff)
            gi, gd = ball_query_raw(nsb, rmax, rmin, xyz.to(d), off.to(d), new_xyz.to(d), noff.to(d))
            if not (torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)):
                print("BALL MISMATCH", tag, nsb, rmax, rmin)
                return 1
            order = ref.make_random_order(off, generator=torch.Generator().manual_seed(seed))
            wi, wd = ref.random_ball_query_raw(nsb, rmax, rmin, xyz, off, new_xyz, noff, order)
            gi, gd = random_ball_query_raw(nsb, rmax, rmin, xyz.to(d), off.to(d), new_xyz.to(d), noff.to(d), order.to(d))
            if not (torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)):
                print("RANDOM BALL MISMATCH", tag, nsb, rmax, rmin)
                return 1
        n_cases += 1
        seed += 1
    where = "host wave64 model%s" % ((", WAVESIM_ORDER=" + os.environ["WAVESIM_ORDER"]) if os.environ.get("WAVESIM_ORDER") else "") if a.model else "cuda:0"
    print(f"fuzz ok on {where}: {n_cases} random layouts (seeds {a.seed0}..{seed - 1}), FPS / kNN / ball / random-ball bit-exact vs the oracle")
    return 0


if __name__ == "__main__":
    sys.exit(main())
