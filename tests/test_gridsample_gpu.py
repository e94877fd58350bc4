"""GPU-side GridSamplePCD (csrc/voxel.hip + bc/gpu_transforms.py) against the reference-generated fixture and the NumPy
oracle: keys / grid coordinates bit-exact, voxel sets identical, injected picks identical."""
import numpy as np
import pytest
import torch

from tests.test_gridsample_cpu import GOLD, rows_as_set

pytestmark = pytest.mark.gpu
DEV = "cuda"


def packed(fx):
    coords = [fx[f"{i}.coord"] for i in range(3)]
    colors = [fx[f"{i}.color"] for i in range(3)]
    sizes = [c.shape[0] for c in coords]
    off = np.cumsum(sizes)
    return (torch.from_numpy(np.concatenate(coords)).to(DEV), torch.from_numpy(np.concatenate(colors)).to(DEV),
            torch.tensor(off, dtype=torch.int64, device=DEV), [0] + off.tolist())


def test_voxel_keys_bit_exact():
    from pointcloudmatters_amd.bc.gpu_transforms import voxel_keys

    fx = np.load(GOLD)
    coord, _, offset, b = packed(fx)
    grid, key, cloud, _ = voxel_keys(coord, offset, 0.005)
    for i in range(3):
        s, e = b[i], b[i + 1]
        assert np.array_equal(grid[s:e].cpu().numpy(), fx[f"{i}.grid_all"])
        assert np.array_equal(key[s:e].cpu().numpy().view(np.uint64), fx[f"{i}.key_all"])
        assert (cloud[s:e] == i).all()


@pytest.mark.parametrize("shuffle", [False, True])
def test_grid_sample_batch_matches_reference_voxels(shuffle):
    from pointcloudmatters_amd.bc.gpu_transforms import grid_sample_batch

    fx = np.load(GOLD)
    coord, color, offset, b = packed(fx)
    g = torch.Generator(device=DEV).manual_seed(7)
    out = grid_sample_batch(coord, offset, {"color": color}, 0.005, shuffle=shuffle, generator=g)
    host = [0] + out["offset"]._pcm_host
    assert out["offset"].tolist() == out["offset"]._pcm_host
    idx = out["index"].cpu().numpy()
    assert len(set(idx.tolist())) == idx.shape[0]
    for i in range(3):
        s, e = host[i], host[i + 1]
        ref_grid = fx[f"{i}.out.grid_coord"]
        assert e - s == ref_grid.shape[0]
        assert ((idx[s:e] >= b[i]) & (idx[s:e] < b[i + 1])).all()  # clouds stay contiguous and in order
        assert rows_as_set(out["grid_coord"][s:e].cpu().numpy()) == rows_as_set(ref_grid)
        # gathered fields belong to the chosen points, and each chosen point lies in the voxel reported for it
        assert np.array_equal(out["coord"][s:e].cpu().numpy(), fx[f"{i}.coord"][idx[s:e] - b[i]])
        assert np.array_equal(out["color"][s:e].cpu().numpy(), fx[f"{i}.color"][idx[s:e] - b[i]])
        assert np.array_equal(out["grid_coord"][s:e].cpu().numpy(), fx[f"{i}.grid_all"][idx[s:e] - b[i]])
    assert int(out["count"].sum()) == coord.shape[0]


def test_grid_sample_batch_injected_pick_equals_oracle():
    """With the random draw injected, the GPU path and the NumPy oracle keep exactly the same points.  The two order the
    voxels differently (uint64 key vs its int64 bit pattern), so the injected numbers are attached to voxels through
    their keys: voxel with key k of cloud i uses draw(i, k)."""
    from oracle import gridsample_cpu as G
    from pointcloudmatters_amd.bc.gpu_transforms import grid_sample_batch

    fx = np.load(GOLD)
    coord, _, offset, b = packed(fx)

    def cloud_of(j):
        return 0 if j < b[1] else (1 if j < b[2] else 2)

    def draw(i, key):
        return (int(key) * 2654435761 + i) % 1000003

    # rand = 0 everywhere: both sides keep the lowest original index of every voxel (stable sorts)
    first = grid_sample_batch(coord, offset, None, 0.005, rand=lambda m: torch.zeros(m, dtype=torch.int64), shuffle=False)
    want0 = np.concatenate([G.grid_sample(fx[f"{i}.coord"], 0.005)[0] + b[i] for i in range(3)])
    assert np.array_equal(np.sort(first["index"].cpu().numpy()), np.sort(want0))
    # arbitrary draws, tied to voxels by key
    keys_gpu = [(cloud_of(j), fx[f"{cloud_of(j)}.key_all"][j - b[cloud_of(j)]]) for j in first["index"].cpu().tolist()]
    inj = torch.tensor([draw(i, k) for i, k in keys_gpu], dtype=torch.int64)
    got = grid_sample_batch(coord, offset, None, 0.005, rand=inj, shuffle=False)["index"].cpu().numpy()
    want = []
    for i in range(3):
        key = G.grid_sample(fx[f"{i}.coord"], 0.005)[2]
        r = np.array([draw(i, k) for k in np.unique(key)], dtype=np.int64)  # oracle order = ascending uint64 key
        want.append(G.grid_sample(fx[f"{i}.coord"], 0.005, rand=r)[0] + b[i])
    assert np.array_equal(np.sort(got), np.sort(np.concatenate(want)))
    assert not np.array_equal(np.sort(got), np.sort(want0))  # the draws really changed the picks


def test_pipeline_output_feeds_the_policy_layout():
    from pointcloudmatters_amd.bc.gpu_transforms import GpuPcdPipeline

    fx = np.load(GOLD)
    coord, color, offset, _ = packed(fx)
    pcds = GpuPcdPipeline(0.005)(coord, color, offset, generator=torch.Generator(device=DEV).manual_seed(1))
    m = pcds["coord"].shape[0]
    assert pcds["feat"].shape == (m, 6) and pcds["grid_coord"].shape == (m, 3) and pcds["offset"][-1].item() == m
    assert pcds["feat"][:, :3].min() >= -1 and pcds["feat"][:, :3].max() <= 1  # NormalizeColorPCD: c / 127.5 - 1
    assert torch.equal(pcds["feat"][:, 3:], pcds["coord"])
    # empty input and a single point
    from pointcloudmatters_amd.bc.gpu_transforms import grid_sample_batch
    one = grid_sample_batch(coord[:1].contiguous(), torch.tensor([1], device=DEV), None, 0.005)
    assert one["index"].tolist() == [0] and one["offset"].tolist() == [1]
