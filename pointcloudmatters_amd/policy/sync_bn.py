"""Synchronised BatchNorm for the flat / hybrid trainer modes (``sync_batchnorm: true`` of configs/trainer/ddp.yaml:9).

The BatchNorm layers on the hot path are owned by fused HIP kernels (policy/bn_relu.py: PointNet's Linear -> BN -> ReLU;
policy/sa_fused.py: the set-abstraction layer's BN): those kernels produce LOCAL sums, the statistics of all ranks are
combined here with the same algebra as torch's SyncBatchNorm --

    forward : all_gather [mean_r, M2_r, n_r]  ->  Chan's parallel combination  ->  global mean / biased variance
    backward: all_reduce [sum dy, sum dy * xhat]  ->  the input gradient uses the global sums and the global count;
              the weight / bias gradients stay local (the gradient exchange averages them like every other parameter)

-- and handed back to the kernels' apply stages.  Every other BatchNorm module becomes a ``torch.nn.SyncBatchNorm``.
The collectives are plain eager calls: in hybrid mode the tokenizer (the only part with BatchNorm) runs outside the
captured graphs."""
import torch
import torch.distributed as dist
import torch.nn as nn


def enable_sync_batchnorm(policy, process_group=None):
    """Mark the BatchNorm1d layers the fused kernels own (`policy.fused_batchnorms()`) for statistic exchange and convert
    every other BatchNorm module to torch's SyncBatchNorm in place."""
    fused = set()
    for owner in ([policy] + [m for m in policy.modules() if m is not policy]):
        get = getattr(owner, "fused_batchnorms", None)
        if get is not None:
            fused.update(id(m) for m in get())
    for m in policy.modules():
        if id(m) in fused:
            m._pcm_sync = process_group if process_group is not None else True

    def convert(module):
        for name, child in list(module.named_children()):
            if isinstance(child, nn.modules.batchnorm._BatchNorm) and id(child) not in fused:
                setattr(module, name, nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))
            else:
                convert(child)

    convert(policy)
    return policy


def multi_rank():
    """A process group with more than one rank is up.  PCM_DP_SINGLE_RANK=1 (tools/dbg/dp_single_rank.py) also accepts a ONE-rank
    group: every collective of the data-parallel path is then really issued -- RCCL launch, stream hand-over and all -- on a
    single GPU, which is how the host / launch overhead of the N > 1 program is measured on a one-GPU box."""
    import os

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PCM_DP_SINGLE_RANK") == "1"


def wants_sync(bn):
    return bool(getattr(bn, "_pcm_sync", False)) and multi_rank()


def _group(bn):
    g = getattr(bn, "_pcm_sync", None)
    return None if g is True else g


def combine_forward(bn, mean_loc, m2_loc, count_loc):
    """Local per-channel mean (C) and sum of squared deviations M2 (C) over `count_loc` rows -> stat (4, C) =
    {mean, invstd, a = gamma * invstd, b = beta - a * mean} of the GLOBAL batch, and n_loc / N; updates the running
    statistics like BatchNorm (momentum, unbiased variance) -- identically on every rank."""
    C = mean_loc.shape[0]
    pack = torch.cat([mean_loc.float(), m2_loc.float(), mean_loc.new_full((1,), float(count_loc), dtype=torch.float32)])
    world = dist.get_world_size(_group(bn))
    parts = [torch.empty_like(pack) for _ in range(world)]
    from .. import _graphs

    grp = _group(bn)
    _graphs.between(lambda: dist.all_gather(parts, pack, group=grp))
    allp = torch.stack(parts)
    n_r = allp[:, 2 * C:].double()                    # (W, 1)
    mean_r, m2_r = allp[:, :C].double(), allp[:, C: 2 * C].double()
    n = n_r.sum()
    mean = (mean_r * n_r).sum(0) / n
    m2 = (m2_r + n_r * (mean_r - mean) ** 2).sum(0)
    var = (m2 / n).clamp_min(0.0)
    invstd = torch.rsqrt(var + bn.eps)
    a = bn.weight.double() * invstd
    stat = torch.stack([mean, invstd, a, bn.bias.double() - a * mean]).float().contiguous()
    if bn.track_running_stats and bn.momentum is not None:
        with torch.no_grad():
            mom = bn.momentum
            unbiased = var * (n / (n - 1.0).clamp_min(1.0))
            bn.running_mean.mul_(1.0 - mom).add_(mean.float(), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(unbiased.float(), alpha=mom)
    # the kernels divide by the LOCAL row count (a host scalar); n_loc / N as a device scalar turns that into 1 / N
    # without reading the global count back to the host
    ratio = (float(count_loc) / n).float()
    return stat, ratio


_INTO_TENSOR_OK = {}


def _all_gather_rows(out, row, group):
    """out[r] = row of rank r.  all_gather_into_tensor writes straight into `out` (no staging copies on RCCL); a backend without
    it (decided once per backend) gets the list form on views of `out`.  Inside a segmented capture (_graphs.SegmentedCapture)
    the call is cut out of the graphs and re-issued eagerly between them at every replay."""
    from .. import _graphs

    key = dist.get_backend(group)

    def gather():
        if _INTO_TENSOR_OK.get(key, True):
            try:
                dist.all_gather_into_tensor(out.view(-1), row, group=group)
                _INTO_TENSOR_OK[key] = True
                return
            except (RuntimeError, NotImplementedError):
                _INTO_TENSOR_OK[key] = False
        dist.all_gather(list(out.unbind(0)), row, group=group)

    _graphs.between(gather)


def combine_forward_sums(bn, sums, src, count_loc, row_index=None):
    """combine_forward from the forward kernels' shifted sums (sums (2, C): sum (y - shift), sum (y - shift)^2; shift = row 0 of
    the (rows, C) fp32 / bf16 matrix `src`, or its row *row_index (a device int32; negative: zeros)):
    two launches around the all_gather (csrc/bnrelu.hip pcm_bn_sync_pack / _combine) instead of ~28 one-element framework
    launches per layer -- at N > 1 the eager tokenizer is paced by the host, and six layers of that were ~1 ms per step."""
    from .. import _lib
    from .._lib import raw_stream

    C = int(sums.shape[1])
    dev = sums.device
    L = _lib.load()
    world = dist.get_world_size(_group(bn))
    with torch.cuda.device(dev):
        pack = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
        rc = L.pcm_bn_sync_pack_hip(C, float(count_loc), sums.data_ptr(), src.data_ptr(), int(src.dtype == torch.bfloat16),
                                    row_index.data_ptr() if row_index is not None else 0, pack.data_ptr(), raw_stream())
        _lib.check(rc, "pcm_bn_sync_pack_hip")
        gathered = torch.empty(world, 2 * C + 1, dtype=torch.float32, device=dev)
        _all_gather_rows(gathered, pack, _group(bn))
        stat = torch.empty(4, C, dtype=torch.float32, device=dev)
        ratio = torch.empty((), dtype=torch.float32, device=dev)
        track = bn.track_running_stats and bn.momentum is not None
        rc = L.pcm_bn_sync_combine_hip(world, C, gathered.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps),
                                       float(bn.momentum if bn.momentum is not None else 0.0),
                                       bn.running_mean.data_ptr() if track else 0, bn.running_var.data_ptr() if track else 0,
                                       float(count_loc), stat.data_ptr(), ratio.data_ptr(), raw_stream())
        _lib.check(rc, "pcm_bn_sync_combine_hip")
    return stat, ratio


def reduce_backward(bn, sums_loc, ratio):
    """(2, C) local {sum dy, sum dy * xhat} -> the same sums over all ranks, pre-scaled by n_loc / N (see combine_forward);
    the local copy stays untouched (it is the weight / bias gradient)."""
    from .. import _graphs

    g = sums_loc.clone()
    grp = _group(bn)
    _graphs.between(lambda: dist.all_reduce(g, group=grp))  # eager between two graph segments when a step is being captured
    return g * ratio
