"""Seeded synthetic clouds in the packed (n,3)+offset layout."""
import numpy as np
import torch


def make_clouds(sizes, seed=0, mode="uniform", dup_frac=0.05, lattice=0.005):
    """mode: 'uniform' table-top extent; 'lattice' snaps to a grid (forces exact distance ties);
    'dup' additionally duplicates dup_frac of the points."""
    rng = np.random.default_rng(seed)
    parts = []
    for n in sizes:
        p = np.empty((n, 3), dtype=np.float32)
        p[:, 0:2] = rng.uniform(-0.4, 0.4, (n, 2))
        p[:, 2] = rng.uniform(0.005, 0.4, n)
        if mode in ("lattice", "dup"):
            p = (np.round(p / lattice) * lattice).astype(np.float32)
        if mode == "dup" and n > 1:
            k = max(1, int(n * dup_frac))
            src = rng.integers(0, n, k)
            dst = rng.integers(0, n, k)
            p[dst] = p[src]
        parts.append(p.astype(np.float32))
    xyz = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
    offset = np.cumsum(sizes).astype(np.int32)
    return torch.from_numpy(xyz).contiguous(), torch.from_numpy(offset)


def new_offsets(ms):
    return torch.from_numpy(np.cumsum(ms).astype(np.int32))
