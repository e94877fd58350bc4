"""Synthetic BC batches with the reference's collated layout.

``pcds = {coord (n,3) f32, grid_coord (n,3) i64, feat (n,6) f32 = [color, coord], offset (b) i64}``,
``qpos (B,9)``, ``actions (B,100,7)``, ``is_pad (B,100) bool``, ``goal_cond (B,3)`` -- what
pcd_collate_fn (/root/reference/src/utils/sparse_tensor_utils.py:65-82) hands to training_step.
Offsets are built on the host first and carry their host copy to the device (no sync later).
"""
import numpy as np
import torch


def make_act_batch(batch, n_points, seed=1000, ragged=False, num_queries=100, action_dim=7, qpos_dim=9,
                   goal_cond_dim=3, device="cpu", grid_size=0.005, feat_keys=("color", "coord")):
    """feat_keys: what CollectPCD concatenates into `feat` -- ("color", "coord") for the default configs
    (configs/data/maniskill2_act_pcd_dataset.yaml:32-34), ("coord",) for the `_wo_rgb` and ("color",) for the `_wo_xyz`
    experiment variants (e.g. scratch_pointnet_pcd_presample_wo_rgb.yaml:18-31)."""
    rng = np.random.default_rng(seed)
    if ragged:
        sizes = rng.integers(int(0.75 * n_points), int(1.25 * n_points) + 1, size=batch).tolist()
    else:
        sizes = [n_points] * batch
    n = int(sum(sizes))
    coord = np.empty((n, 3), dtype=np.float32)
    coord[:, 0:2] = rng.uniform(-0.4, 0.4, (n, 2))
    coord[:, 2] = rng.uniform(0.005, 0.4, n)  # table-top extent after the z > 0.005 cut
    color = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)  # after NormalizeColorPCD
    offset_host = np.cumsum(sizes).tolist()
    grid = np.floor(coord / grid_size).astype(np.int64)
    start = 0
    for e in offset_host:  # GridSamplePCD: grid_coord is relative to the cloud's min corner
        grid[start:e] -= grid[start:e].min(axis=0, keepdims=True)
        start = e
    is_pad = np.zeros((batch, num_queries), dtype=bool)
    for i, t in enumerate(rng.integers(0, 51, size=batch)):
        if t:
            is_pad[i, num_queries - t :] = True
    dev = torch.device(device)
    offset = torch.tensor(offset_host, dtype=torch.int64).to(dev)
    offset._pcm_host = [int(v) for v in offset_host]
    coord_t = torch.from_numpy(coord).to(dev)
    batch_dict = {
        "pcds": {
            "coord": coord_t,
            "grid_coord": torch.from_numpy(grid).to(dev),
            "feat": torch.cat([{"color": torch.from_numpy(color).to(dev), "coord": coord_t}[k] for k in feat_keys], dim=1).contiguous(),
            "offset": offset,
        },
        "qpos": torch.from_numpy(rng.standard_normal((batch, qpos_dim)).astype(np.float32)).to(dev),
        "actions": torch.from_numpy(rng.standard_normal((batch, num_queries, action_dim)).astype(np.float32)).to(dev),
        "is_pad": torch.from_numpy(is_pad).to(dev),
    }
    if goal_cond_dim > 0:
        batch_dict["goal_cond"] = torch.from_numpy(rng.standard_normal((batch, goal_cond_dim)).astype(np.float32)).to(dev)
    return batch_dict


def make_dp_batch(batch, n_points, seed=1000, ragged=False, horizon=16, n_obs_steps=2, action_dim=7, qpos_dim=9,
                  device="cpu", goal_dim=0, feat_keys=("color", "coord")):
    """Diffusion-Policy batch: B samples, each with n_obs_steps clouds flattened sample-major
    (sparse_tensor_utils.py:74-75 -> b = B*To clouds), qpos window (B, horizon, qpos_dim), action (B, horizon, Da)."""
    clouds = make_act_batch(batch * n_obs_steps, n_points, seed=seed, ragged=ragged, num_queries=1, action_dim=1,
                            qpos_dim=1, goal_cond_dim=0, device=device, feat_keys=feat_keys)["pcds"]
    rng = np.random.default_rng(seed + 1)
    dev = torch.device(device)
    out = {
        "obs": {"pcds": clouds,
                "qpos": torch.from_numpy(rng.uniform(-1, 1, (batch, horizon, qpos_dim)).astype(np.float32)).to(dev)},
        "action": torch.from_numpy(rng.uniform(-1, 1, (batch, horizon, action_dim)).astype(np.float32)).to(dev),
    }
    if goal_dim:  # RLBench: a language embedding of the task (rlbench_diffusion_policy_model.yaml:26-28)
        out["goal"] = {"task_emb": torch.from_numpy(rng.standard_normal((batch, goal_dim)).astype(np.float32)).to(dev)}
    return out


def clone_batch(batch):
    """Shallow per-step copy: the policy writes intermediate results into the dict it is given."""
    out = {}
    for k, v in batch.items():
        out[k] = clone_batch(v) if isinstance(v, dict) else v
    return out
