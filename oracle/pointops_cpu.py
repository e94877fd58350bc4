"""CPU ORACLE: the reference's ``pointops`` Python API on CPU tensors (test infrastructure).

Mirrors ``/root/reference/libs/pointops/functions/__init__.py:1-14`` name for name.  Each wrapper
follows the allocation / pre-fill / post-processing the reference wrapper does around its native
call (cited per function) with ``torch.cuda.*Tensor`` replaced by CPU tensors and
``pointops._C.*_cuda`` replaced by the C oracle ``pcm_*_cpu`` (oracle/pcm_oracle.c).
"""
import torch
from torch.autograd import Function

from . import lib as _lib


def _p(t):
    return t.data_ptr()


def _i32(t):
    return t.to(torch.int32).contiguous()


def _f32c(t):
    assert t.dtype == torch.float32, t.dtype
    return t.contiguous()


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} returned {rc}")


# ----------------------------------------------------------------------------- sampling.py:6-26
def farthest_point_sampling(xyz, offset, new_offset):
    assert xyz.is_contiguous() and not xyz.is_cuda
    L = _lib.load()
    n, b = xyz.shape[0], offset.shape[0]
    off = [int(v) for v in offset.tolist()]
    n_max = off[0]
    for i in range(1, b):
        n_max = max(off[i] - off[i - 1], n_max)
    m = int(new_offset[b - 1].item())
    idx = torch.zeros(m, dtype=torch.int32)
    tmp = torch.full((n,), 1e10, dtype=torch.float32)
    o32, no32 = _i32(offset), _i32(new_offset)
    _check(
        L.pcm_farthest_point_sampling_cpu(b, n_max, _p(xyz), _p(o32), _p(no32), _p(tmp), _p(idx)),
        "fps",
    )
    return idx


# ----------------------------------------------------------------------------- query.py:6-112
def knn_query_raw(nsample, xyz, offset, new_xyz=None, new_offset=None):
    """Returns (idx, dist2) -- the native outputs before the wrapper's sqrt (query.py:23)."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    L = _lib.load()
    m = new_xyz.shape[0]
    idx = torch.zeros(m, nsample, dtype=torch.int32)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32)
    o32, no32 = _i32(offset), _i32(new_offset)
    _check(
        L.pcm_knn_query_cpu(m, nsample, _p(xyz), _p(new_xyz), _p(o32), _p(no32), _p(idx), _p(dist2)),
        "knn_query",
    )
    return idx, dist2


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    idx, dist2 = knn_query_raw(nsample, xyz, offset, new_xyz, new_offset)
    return idx, torch.sqrt(dist2)


def ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    assert min_radius < max_radius
    L = _lib.load()
    m = new_xyz.shape[0]
    idx = torch.zeros(m, nsample, dtype=torch.int32)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32)
    o32, no32 = _i32(offset), _i32(new_offset)
    _check(
        L.pcm_ball_query_cpu(
            m, nsample, min_radius, max_radius, _p(xyz), _p(new_xyz), _p(o32), _p(no32), _p(idx), _p(dist2)
        ),
        "ball_query",
    )
    return idx, dist2


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    idx, dist2 = ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset)
    return idx, torch.sqrt(dist2)


def make_random_order(offset, generator=None):
    """query.py:46-53: per-cloud randperm + start, concatenated (int32)."""
    order, start = [], 0
    for e in [int(v) for v in offset.tolist()]:
        order.append(torch.randperm(e - start, dtype=torch.int32, generator=generator) + start)
        start = e
    return torch.cat(order, dim=0)


def random_ball_query_raw(
    nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None
):
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    assert min_radius < max_radius
    L = _lib.load()
    m = new_xyz.shape[0]
    if order is None:
        order = make_random_order(offset)
    order = _i32(order)
    idx = torch.zeros(m, nsample, dtype=torch.int32)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32)
    o32, no32 = _i32(offset), _i32(new_offset)
    _check(
        L.pcm_random_ball_query_cpu(
            m, nsample, min_radius, max_radius, _p(order), _p(xyz), _p(new_xyz), _p(o32), _p(no32),
            _p(idx), _p(dist2),
        ),
        "random_ball_query",
    )
    return idx, dist2


def random_ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None):
    idx, dist2 = random_ball_query_raw(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset, order)
    return idx, torch.sqrt(dist2)


# ----------------------------------------------------------------------------- grouping.py:6-62
class _Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        assert input.is_contiguous() and idx.is_contiguous()
        L = _lib.load()
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.empty(m, nsample, c, dtype=torch.float32)
        _check(L.pcm_grouping_forward_cpu(m, nsample, c, _p(input), _p(idx), _p(output)), "grouping_fwd")
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        (idx,) = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        m, nsample, c = grad_output.shape
        grad_input = torch.zeros(ctx.n, c, dtype=torch.float32)
        _check(
            L.pcm_grouping_backward_cpu(m, nsample, c, _p(grad_output), _p(idx), _p(grad_input)),
            "grouping_bwd",
        )
        return grad_input, None


grouping2 = _Grouping.apply


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """grouping.py:35-59 restated: zero row for idx == -1, relative xyz masked by sign(idx+1),
    xyz channels first."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
    flat = idx.reshape(-1).long()
    feat_pad = torch.cat([feat, feat.new_zeros(1, c)], dim=0)
    grouped_feat = feat_pad[flat].view(m, nsample, c)
    if not with_xyz:
        return grouped_feat
    assert new_xyz.is_contiguous()
    xyz_pad = torch.cat([xyz, xyz.new_zeros(1, 3)], dim=0)
    valid = torch.sign(idx + 1).to(xyz.dtype)  # 0 where idx == -1, 1 otherwise
    rel = xyz_pad[flat].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    rel = rel * valid.unsqueeze(-1)
    return torch.cat((rel, grouped_feat), dim=-1)


# ----------------------------------------------------------------------------- interpolation.py
def _interp_weights(xyz, new_xyz, offset, new_offset, k):
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    return idx, dist_recip / norm


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """interpolation.py:8-22 (pure torch accumulation in k order)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, weight = _interp_weights(xyz, new_xyz, offset, new_offset, k)
    new_feat = torch.zeros(new_xyz.shape[0], feat.shape[1], dtype=torch.float32)
    for i in range(k):
        new_feat = new_feat + feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return new_feat


class _Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        L = _lib.load()
        idx, weight = _interp_weights(xyz, new_xyz, offset, new_offset, k)
        weight = weight.contiguous()
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        output = torch.zeros(n, c, dtype=torch.float32)
        _check(
            L.pcm_interpolation_forward_cpu(n, c, k, _p(input), _p(idx), _p(weight), _p(output)),
            "interp_fwd",
        )
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        idx, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        grad_input = torch.zeros(ctx.m, c, dtype=torch.float32)
        _check(
            L.pcm_interpolation_backward_cpu(n, c, ctx.k, _p(grad_output), _p(idx), _p(weight), _p(grad_input)),
            "interp_bwd",
        )
        return None, None, grad_input, None, None, None


interpolation2 = _Interpolation.apply


# ----------------------------------------------------------------------------- subtraction.py
class _Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        assert input1.is_contiguous() and input2.is_contiguous()
        L = _lib.load()
        n, c = input1.shape
        nsample = idx.shape[-1]
        idx = idx.contiguous()
        output = torch.zeros(n, nsample, c, dtype=torch.float32)
        _check(
            L.pcm_subtraction_forward_cpu(n, nsample, c, _p(input1), _p(input2), _p(idx), _p(output)),
            "sub_fwd",
        )
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        (idx,) = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = grad_output.shape
        g1 = torch.zeros(n, c, dtype=torch.float32)
        g2 = torch.zeros(n, c, dtype=torch.float32)
        _check(
            L.pcm_subtraction_backward_cpu(n, nsample, c, _p(idx), _p(grad_output), _p(g1), _p(g2)),
            "sub_bwd",
        )
        return g1, g2, None


subtraction = _Subtraction.apply


# ----------------------------------------------------------------------------- aggregation.py
class _Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        L = _lib.load()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        idx = idx.contiguous()
        output = torch.zeros(n, c, dtype=torch.float32)
        _check(
            L.pcm_aggregation_forward_cpu(
                n, nsample, c, w_c, _p(input), _p(position), _p(weight), _p(idx), _p(output)
            ),
            "agg_fwd",
        )
        ctx.save_for_backward(input, position, weight, idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        input, position, weight, idx = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        gi = torch.zeros(n, c, dtype=torch.float32)
        gp = torch.zeros(n, nsample, c, dtype=torch.float32)
        gw = torch.zeros(n, nsample, w_c, dtype=torch.float32)
        _check(
            L.pcm_aggregation_backward_cpu(
                n, nsample, c, w_c, _p(input), _p(position), _p(weight), _p(idx), _p(grad_output),
                _p(gi), _p(gp), _p(gw),
            ),
            "agg_bwd",
        )
        return gi, gp, gw, None


aggregation = _Aggregation.apply


# ----------------------------------------------------------------------------- attention.py
class _AttentionRelationStep(Function):
    @staticmethod
    def forward(ctx, query, key, weight, index_target, index_refer):
        assert query.is_contiguous() and key.is_contiguous() and weight.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        L = _lib.load()
        _, g, c = query.shape
        m = index_target.shape[0]
        it, ir = _i32(index_target), _i32(index_refer)
        output = torch.zeros(m, g, dtype=torch.float32)
        _check(
            L.pcm_attention_relation_step_forward_cpu(
                m, g, c, _p(query), _p(key), _p(weight), _p(it), _p(ir), _p(output)
            ),
            "attn_rel_fwd",
        )
        ctx.save_for_backward(query, key, weight, it, ir)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        query, key, weight, it, ir = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, g, c = query.shape
        m = it.shape[0]
        gq = torch.zeros(n, g, c, dtype=torch.float32)
        gk = torch.zeros(n, g, c, dtype=torch.float32)
        gw = torch.zeros(c, dtype=torch.float32)
        _check(
            L.pcm_attention_relation_step_backward_cpu(
                m, g, c, _p(query), _p(gq), _p(key), _p(gk), _p(weight), _p(gw), _p(it), _p(ir),
                _p(grad_output),
            ),
            "attn_rel_bwd",
        )
        return gq, gk, None, None, None  # attention.py:61 returns None for grad_weight


class _AttentionFusionStep(Function):
    @staticmethod
    def forward(ctx, weight, value, index_target, index_refer):
        assert weight.is_contiguous() and value.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        L = _lib.load()
        n, g, c = value.shape
        m = index_refer.shape[0]
        it, ir = _i32(index_target), _i32(index_refer)
        output = torch.zeros(n, g, c, dtype=torch.float32)
        _check(
            L.pcm_attention_fusion_step_forward_cpu(
                m, g, c, _p(weight), _p(value), _p(it), _p(ir), _p(output)
            ),
            "attn_fus_fwd",
        )
        ctx.save_for_backward(weight, value, it, ir)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        L = _lib.load()
        weight, value, it, ir = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, g, c = value.shape
        m = it.shape[0]
        gw = torch.zeros(m, g, dtype=torch.float32)
        gv = torch.zeros(n, g, c, dtype=torch.float32)
        _check(
            L.pcm_attention_fusion_step_backward_cpu(
                m, g, c, _p(weight), _p(gw), _p(value), _p(gv), _p(it), _p(ir), _p(grad_output)
            ),
            "attn_fus_bwd",
        )
        return gw, gv, None, None


attention_relation_step = _AttentionRelationStep.apply
attention_fusion_step = _AttentionFusionStep.apply


# ----------------------------------------------------------------------------- utils.py:5-121
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
    """utils.py:48-99 (dilated kNN + plain gather, no -1 handling)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        total = 1 + (nsample - 1) * (dilation + 1)
        idx_nd, _ = knn_query(total, xyz, offset, new_xyz, new_offset)
        ends = [int(v) for v in offset.tolist()]
        starts = [0] + ends[:-1]
        nends = [int(v) for v in new_offset.tolist()]
        nstarts = [0] + nends[:-1]
        parts = []
        for i in range(offset.shape[0]):
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
    ends = [int(v) for v in offset.tolist()]
    counts = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
    return torch.repeat_interleave(torch.arange(len(ends)), torch.tensor(counts)).long().to(offset.device)


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).int()
