"""GPU: randomized parity sweep of the index kernels against the oracle -- ragged batches, clouds smaller than
nsample or than the number of samples requested, single-point clouds, duplicates and lattice ties, every
block-size class of FPS (BS = 1 ... 1024), plus empty inputs."""
import numpy as np
import pytest
import torch

from tests.util import make_clouds, new_offsets

pytestmark = pytest.mark.gpu


def _case(rng):
    b = int(rng.integers(1, 7))
    scale = int(rng.choice([3, 40, 300, 1500, 5000]))
    sizes = [int(rng.integers(1, scale + 1)) for _ in range(b)]
    ms = [int(rng.integers(1, min(max(s * 2, 2), 1200) + 1)) for s in sizes]
    mode = str(rng.choice(["uniform", "lattice", "dup"]))
    return sizes, ms, mode


@pytest.mark.parametrize("seed", range(24))
def test_fps_knn_random_layouts(hip_device, seed):
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref
    from pointcloudmatters_amd.pointops.query import knn_query_raw

    rng = np.random.default_rng(1000 + seed)
    sizes, ms, mode = _case(rng)
    xyz, off = make_clouds(sizes, seed=seed, mode=mode, lattice=float(rng.choice([0.005, 0.05, 0.2])))
    noff = new_offsets(ms)
    d = hip_device
    want = ref.farthest_point_sampling(xyz, off, noff)
    got = po.farthest_point_sampling(xyz.to(d), off.to(d), noff.to(d))
    assert torch.equal(got.cpu(), want), (sizes, ms, mode)
    new_xyz = xyz[want.long()].contiguous()
    ns = int(rng.choice([1, 3, 16, 32, 63, 64, 100]))
    wi, wd = ref.knn_query_raw(ns, xyz, off, new_xyz, noff)
    gi, gd = knn_query_raw(ns, xyz.to(d), off.to(d), new_xyz.to(d), noff.to(d))
    assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd), (sizes, ms, mode, ns)


@pytest.mark.parametrize("n_max", [1, 2, 3, 7, 8, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4096, 4097, 8192, 8193, 16384, 16385])
def test_fps_every_block_size_class(hip_device, n_max):
    """BS = opt_n_threads(n_max) decides the reference's tie order; sweep the class boundaries with tie-heavy data."""
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    sizes = [n_max, max(1, n_max // 3)]
    ms = [min(n_max + 2, 96), 17]
    xyz, off = make_clouds(sizes, seed=n_max, mode="lattice", lattice=0.1)  # coarse lattice: many exact ties
    noff = new_offsets(ms)
    want = ref.farthest_point_sampling(xyz, off, noff)
    got = po.farthest_point_sampling(xyz.to(hip_device), off.to(hip_device), noff.to(hip_device))
    assert torch.equal(got.cpu(), want)


def test_empty_inputs(hip_device):
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd.pointops.query import knn_query_raw

    d = hip_device
    xyz, off = make_clouds([10], seed=0)
    # zero queries
    empty_q = torch.zeros(0, 3, device=d)
    gi, gd = knn_query_raw(4, xyz.to(d), off.to(d), empty_q, torch.tensor([0], dtype=torch.int32, device=d))
    assert gi.shape == (0, 4) and gd.shape == (0, 4)
    # zero samples requested
    idx = po.farthest_point_sampling(xyz.to(d), off.to(d), torch.tensor([0], dtype=torch.int32, device=d))
    assert idx.numel() == 0
    # grouping with zero rows
    out = po.grouping(torch.zeros(0, 4, dtype=torch.int32, device=d), torch.zeros(10, 5, device=d), xyz.to(d), empty_q, with_xyz=True)
    assert out.shape == (0, 4, 8)


def test_one_cloud_without_samples_between_others(hip_device):
    """new_offset with a zero-length range: the cloud contributes no samples, neighbours keep their slots."""
    import pointcloudmatters_amd.pointops as po
    from oracle import pointops_cpu as ref

    xyz, off = make_clouds([50, 30, 40], seed=3)
    noff = new_offsets([10, 0, 12])
    want = ref.farthest_point_sampling(xyz, off, noff)
    got = po.farthest_point_sampling(xyz.to(hip_device), off.to(hip_device), noff.to(hip_device))
    assert torch.equal(got.cpu(), want)
