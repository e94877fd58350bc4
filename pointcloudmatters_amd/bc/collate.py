"""Packed-batch collation -- defines the ``(n, 3) + cumulative offset`` layout every pointops call
consumes.  Behavioural counterpart of /root/reference/src/utils/sparse_tensor_utils.py:7-82
(``offset2batch``, ``batch2offset``, ``point_collate_fn``, ``pcd_collate_fn``).

Difference in *how*: offsets are computed on the host during collation anyway, so the host copy is
kept attached to the offset tensor (``_pcm_host``) and travels with ``to_device`` -- the GPU path
never has to read an offset back (the reference does, b+1 times per FPS call).
"""
from collections.abc import Mapping, Sequence

import torch
from torch.utils.data import default_collate


def offset2batch(offset):
    host = [int(v) for v in offset.tolist()]
    counts = [host[0]] + [host[i] - host[i - 1] for i in range(1, len(host))]
    return torch.repeat_interleave(torch.arange(len(host)), torch.tensor(counts)).long().to(offset.device)


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()


def point_collate_fn(batch):
    """Concatenate per-cloud tensors; keys containing "offset" become cumulative ends."""
    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    first = batch[0]
    if isinstance(first, torch.Tensor):
        return torch.cat(list(batch))
    if isinstance(first, str):
        return list(batch)
    if isinstance(first, Sequence):
        for data in batch:
            data.append(torch.tensor([data[0].shape[0]]))
        out = [point_collate_fn(samples) for samples in zip(*batch)]
        out[-1] = torch.cumsum(out[-1], dim=0).int()
        return out
    if isinstance(first, Mapping):
        out = {key: point_collate_fn([d[key] for d in batch]) for key in first}
        for key in out:
            if "offset" in key:
                out[key] = torch.cumsum(out[key], dim=0)
                out[key]._pcm_host = [int(v) for v in out[key].tolist()]
        return out
    return default_collate(batch)


def pcd_collate_fn(batch):
    """Samples carry ``pcds`` = list of per-cloud dicts (1 for ACT, n_obs_steps for Diffusion Policy);
    they are flattened sample-major and packed, everything else goes through default_collate."""
    first = batch[0]
    nested = "obs" in first and "pcds" in first["obs"]
    if "pcds" not in first and not nested:
        return default_collate(batch)
    if nested:
        pcds = [b["obs"].pop("pcds") for b in batch]
    else:
        pcds = [b.pop("pcds") for b in batch]
    out = default_collate(batch)
    packed = point_collate_fn([c for sample in pcds for c in sample])
    if nested:
        out["obs"]["pcds"] = packed
    else:
        out["pcds"] = packed
    return out


def to_device(batch, device, non_blocking=True):
    """Move a collated batch to the GPU, keeping the host copy of every offset tensor attached."""
    if isinstance(batch, Mapping):
        return {k: to_device(v, device, non_blocking) for k, v in batch.items()}
    if torch.is_tensor(batch):
        moved = batch.to(device, non_blocking=non_blocking)
        host = getattr(batch, "_pcm_host", None)
        if host is not None:
            moved._pcm_host = host
        return moved
    return batch
