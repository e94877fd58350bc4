"""GPU-side point-cloud data path: what the reference's CPU workers do per sample before collation, for a whole packed
batch at once.

  GridSamplePCD(grid_size, hash_type="fnv", mode="train", return_grid_coord=True)   transformpcd.py:662-793
  NormalizeColorPCD                                                                  transformpcd.py:83-87
  ShufflePointPCD                                                                    transformpcd.py:796-815
  CollectPCD(keys=(coord, grid_coord), feat_keys=(color, coord)) + pcd_collate_fn    transformpcd.py:10-36,
                                                                                     sparse_tensor_utils.py:36-82

The voxel arithmetic (float64 division, floor, per-cloud minimum, FNV-1a 64) is the HIP kernel pair of csrc/voxel.hip
-- bit-exact with the reference's functions; sort / unique / gather are rocPRIM-backed framework ops.  Which point of a
voxel survives is random in the reference (np.random) and random here (torch generator, or injected `rand`); the SET
of occupied voxels, their grid coordinates and the per-cloud counts are identical.  One host synchronisation per batch
(the number of voxels sizes the outputs), against b+1 per FPS call in the reference's wrappers.
"""
import torch

from .. import _lib
from ..pointops import _common as C


def voxel_keys(coord, offset, grid_size):
    """-> grid_coord (n,3) int64 (relative to each cloud's minimum), key (n) int64 (FNV-1a 64 bit pattern),
    cloud (n) int32, gmin (b,3) int32."""
    C.require_hip(coord, offset)
    coord = C.f32c(coord, "coord")
    o32 = C.i32c(offset)
    n, b = int(coord.shape[0]), int(o32.shape[0])
    dev = coord.device
    with torch.cuda.device(dev):
        gmin = torch.empty(b, 3, dtype=torch.int32, device=dev)
        grid = torch.empty(n, 3, dtype=torch.int64, device=dev)
        key = torch.empty(n, dtype=torch.int64, device=dev)
        cloud = torch.empty(n, dtype=torch.int32, device=dev)
        rc = _lib.load().pcm_voxel_keys_hip(n, b, coord.data_ptr(), o32.data_ptr(), float(grid_size), gmin.data_ptr(),
                                            grid.data_ptr(), key.data_ptr(), cloud.data_ptr(), C.stream())
    _lib.check(rc, "pcm_voxel_keys_hip")
    return grid, key, cloud, gmin


def grid_sample_batch(coord, offset, fields=None, grid_size=0.005, rand=None, shuffle=True, generator=None):
    """One random point per occupied voxel, per cloud (GridSamplePCD mode="train"), optionally shuffled within each cloud.

    coord (n,3) f32 on the GPU, offset (b) cumulative ends; `fields` {name: (n, c)} are gathered alongside.
    rand (optional, int64 per voxel in sorted (cloud, key) order, or a callable m -> tensor) replaces the random draw.
    Returns {"index" (m) into the input, "coord", "grid_coord" (m,3) int64, "offset" (b) int64 with its host copy,
    "count" (m) points per surviving voxel, **fields}."""
    grid, key, cloud, _ = voxel_keys(coord, offset, grid_size)
    n, b = coord.shape[0], offset.shape[0]
    dev = coord.device
    # order by (cloud, key): two stable radix sorts (clouds are already contiguous, so the second one is cheap to verify)
    _, p1 = torch.sort(key, stable=True)
    _, p2 = torch.sort(cloud[p1], stable=True)
    idx_sort = p1[p2]
    key_s, cloud_s = key[idx_sort], cloud[idx_sort]
    flag = torch.ones(n, dtype=torch.bool, device=dev)
    if n > 1:
        flag[1:] = (key_s[1:] != key_s[:-1]) | (cloud_s[1:] != cloud_s[:-1])
    starts = torch.nonzero(flag).squeeze(1)  # the one host sync: number of voxels
    m = int(starts.shape[0])
    count = torch.diff(starts, append=torch.tensor([n], device=dev))
    vcloud = cloud_s[starts].long()
    per_cloud = torch.bincount(vcloud, minlength=b)
    if rand is None:
        # np.random.randint(0, count.max(), count.size) % count, with count.max() taken per cloud
        cmax = torch.zeros(b, dtype=torch.int64, device=dev).scatter_reduce_(0, vcloud, count, reduce="amax")[vcloud]
        u = torch.rand(m, device=dev, generator=generator, dtype=torch.float64)
        r = torch.minimum((u * cmax).long(), cmax - 1) % count
    else:
        r = (rand(m) if callable(rand) else rand).to(dev).long() % count
    index = idx_sort[starts + r]
    if shuffle:
        noise = torch.rand(m, device=dev, generator=generator, dtype=torch.float64)
        order = torch.argsort(vcloud.double() + noise)  # clouds stay contiguous and in order
        index, count = index[order], count[order]
    host = torch.cumsum(per_cloud, 0).tolist()
    new_offset = C.with_host(torch.tensor(host, dtype=torch.int64, device=dev), host)
    out = {"index": index, "coord": coord[index], "grid_coord": grid[index], "offset": new_offset, "count": count}
    for k, v in (fields or {}).items():
        out[k] = v[index]
    return out


class GpuPcdPipeline:
    """GridSamplePCD -> NormalizeColorPCD -> ShufflePointPCD -> CollectPCD + collate, on the GPU, for a packed batch.
    Input: coord (n,3) f32, color (n,3) (0..255), offset (b).  Output: the `pcds` dict the policies consume."""

    def __init__(self, grid_size=0.005, normalize_color=True, shuffle=True):
        self.grid_size, self.normalize_color, self.shuffle = grid_size, normalize_color, shuffle

    def __call__(self, coord, color, offset, generator=None):
        s = grid_sample_batch(coord, offset, {"color": color}, self.grid_size, shuffle=self.shuffle, generator=generator)
        col = s["color"].float()
        if self.normalize_color:
            col = col / 127.5 - 1
        return {"coord": s["coord"], "grid_coord": s["grid_coord"], "feat": torch.cat([col, s["coord"]], dim=1),
                "offset": s["offset"]}
