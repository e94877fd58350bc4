"""NumPy restatement of the reference's GridSamplePCD arithmetic (train mode).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/data/components/transformpcd.py:684-705 (scale, floor, per-cloud minimum, key, argsort,
unique, pick) and :776-790 (fnv_hash_vec).  Pinned against the reference's own functions, run in the build container
with NumPy 2.2.6, by tests/golden/gridsample_ref.npz (keys and grid coordinates bit-exact, voxel sets identical).
The random pick is an input (`rand`, one integer per voxel in sorted-key order) so that it can be injected.
"""
import numpy as np


def fnv_hash_vec(arr):
    arr = arr.astype(np.uint64)
    h = np.full(arr.shape[0], 14695981039346656037, dtype=np.uint64)
    for j in range(arr.shape[1]):
        h = h * np.uint64(1099511628211)
        h = np.bitwise_xor(h, arr[:, j])
    return h


def voxel_keys(coord, grid_size):
    scaled = coord / np.array(grid_size)  # float32 array / float64 0-d array -> float64 under NumPy >= 2
    grid = np.floor(scaled).astype(np.int64)
    gmin = grid.min(0)
    grid = grid - gmin
    return grid, fnv_hash_vec(grid), gmin


def grid_sample(coord, grid_size, rand=None):
    """-> (index of the chosen point per voxel in sorted-key order, grid (n,3), key (n), count per voxel).  Ties inside a
    voxel are ordered by original index (stable sort); rand=None picks the first."""
    grid, key, _ = voxel_keys(coord, grid_size)
    idx_sort = np.argsort(key, kind="stable")
    _, count = np.unique(key[idx_sort], return_counts=True)
    starts = np.cumsum(np.insert(count, 0, 0)[:-1])
    r = np.zeros(count.size, dtype=np.int64) if rand is None else np.asarray(rand) % count
    return idx_sort[starts + r], grid, key, count
