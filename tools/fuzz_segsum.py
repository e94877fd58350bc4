#!/usr/bin/env python
"""Randomized campaign for csrc/segsum.hip on the GPU box: scatter plan == numpy CSR, planned / implicit segment sums ==
float64 index arithmetic, over random row counts, destination counts, widths, strides, offsets, holes and hubs."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pointcloudmatters_amd.pointops import _common as C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60)
args = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
t0, rounds = time.time(), 0
while time.time() - t0 < args.seconds:
    rows = int(rng.choice([0, 1, 7, 100, 3000, 40000, 250000]))
    rows = int(rng.integers(0, rows + 1)) if rows else 0
    n_dst = int(rng.choice([1, 2, 33, 1000, 4097, 70000]))
    c = int(rng.choice([1, 3, 4, 5, 32, 96, 128, 260, 512]))
    off = int(rng.choice([0, 0, 3, 4]))
    stride = c + off + int(rng.choice([0, 0, 1, 4]))
    idx = rng.integers(0, n_dst, rows).astype(np.int32)
    if rows and rng.random() < 0.5:
        idx[rng.random(rows) < 0.1] = -1
    if rows > 10 and rng.random() < 0.3:
        idx[: rows // 2] = int(rng.integers(0, n_dst))
    mode = int(rng.integers(0, 3))
    w_c = int(rng.choice([1, 2, 4, 8])) if mode == 2 else 1
    if mode == 2 and c % w_c:
        w_c = 1
    rowdiv = int(rng.choice([1, 1, 3]))
    src_rows = (rows + rowdiv - 1) // rowdiv if rows else 0
    src = rng.standard_normal((max(src_rows, 1), stride)).astype(np.float32)
    scale = rng.random((max(rows, 1), w_c)).astype(np.float32) if mode else None
    want = np.zeros((n_dst, c), np.float64)
    for r in range(rows) if rows <= 3000 else []:
        j = idx[r]
        if j >= 0:
            s = 1.0 if mode == 0 else (scale[r, 0] if mode == 1 else scale[r][np.arange(c) % w_c])
            want[j] += src[r // rowdiv, off:off + c].astype(np.float64) * s
    if rows > 3000:
        keep = idx >= 0
        sc = np.ones((rows, c)) if mode == 0 else (np.repeat(scale[:rows, :1], c, 1) if mode == 1 else scale[:rows][:, np.arange(c) % w_c])
        np.add.at(want, idx[keep], src[np.arange(rows) // rowdiv][keep][:, off:off + c].astype(np.float64) * sc[keep])
    ti = torch.from_numpy(idx).to(dev)
    plan = C.ScatterPlan(ti, n_dst)
    dst = torch.full((n_dst, c), float("nan"), device=dev)
    C.segment_sum(dst, torch.from_numpy(src).to(dev), src_stride=stride, src_off=off, plan=plan, rowdiv=rowdiv,
                  scale=torch.from_numpy(scale[:max(rows, 1)].reshape(-1) if mode == 1 else scale).to(dev) if mode else None,
                  scale_mode=mode, w_c=w_c, sign=-1.0 if rounds % 2 else 1.0)
    got = dst.cpu().numpy().astype(np.float64) * (-1.0 if rounds % 2 else 1.0)
    tol = 1e-5 * max(1.0, np.abs(want).max()) * max(1, int(np.sqrt(max(1, np.bincount(idx[idx >= 0], minlength=1).max() if rows else 1))))
    assert np.isfinite(got).all() and np.abs(got - want).max() <= tol, (rounds, rows, n_dst, c, stride, off, mode, w_c, rowdiv, np.abs(got - want).max(), tol)
    rounds += 1
print("fuzz ok: %d random segment-sum layouts in %.0f s" % (rounds, time.time() - t0))
