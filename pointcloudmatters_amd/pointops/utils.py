"""*_and_group helpers, offset2batch, batch2offset -- mirrors
/root/reference/libs/pointops/functions/utils.py:5-121."""
import torch

from . import _common as C
from .grouping import grouping
from .query import ball_query, knn_query


def knn_query_and_group(
    feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None, with_xyz=False
):
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def ball_query_and_group(
    feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, max_radio=None, min_radio=0,
    nsample=None, with_xyz=False,
):
    if idx is None:
        assert nsample is not None and offset is not None
        assert max_radio is not None and min_radio is not None
        idx, _ = ball_query(nsample, max_radio, min_radio, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def query_and_group(
    nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, with_feat=True, with_xyz=True
):
    """utils.py:48-99: dilated kNN neighbourhoods, plain gather (no -1 handling)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        total = 1 + (nsample - 1) * (dilation + 1)
        idx_nd, _ = knn_query(total, xyz, offset, new_xyz, new_offset)
        ends = C.host_offsets(offset)
        starts = [0] + ends[:-1]
        nends = C.host_offsets(new_offset)
        nstarts = [0] + nends[:-1]
        parts = []
        for i in range(len(ends)):
            if ends[i] - starts[i] < total:
                soft = (ends[i] - starts[i] - 1) / (nsample - 1) - 1
            else:
                soft = dilation
            cols = [int((soft + 1) * j) for j in range(nsample)]
            parts.append(idx_nd[nstarts[i] : nends[i], cols])
        idx = torch.cat(parts, dim=0)
    if not with_feat:
        return idx
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.reshape(-1).long()
    grouped_xyz = xyz[flat, :].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    grouped_feat = feat[flat, :].view(m, nsample, c)
    if with_xyz:
        return torch.cat((grouped_xyz, grouped_feat), -1), idx
    return grouped_feat, idx


def offset2batch(offset):
    """utils.py:102-117 builds this with a Python loop over device scalars; same result, one sync."""
    counts = torch.tensor(C.counts_from_offsets(C.host_offsets(offset)), device=offset.device)
    return torch.repeat_interleave(torch.arange(counts.numel(), device=offset.device), counts).long()


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).int()
