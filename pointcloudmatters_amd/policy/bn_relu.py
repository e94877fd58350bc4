"""Fused BatchNorm1d + ReLU over packed point features (csrc/bnrelu.hip) as an autograd function.

``bn_relu(y, bn)`` == ``relu(bn(y))`` for y (n, C) on a HIP device: training mode uses batch statistics and updates
the running ones exactly like ``nn.BatchNorm1d``; eval mode applies the running statistics.  Output dtype = input
dtype (bf16 under autocast, as torch's batch_norm).  Host tensors and unsupported layouts take the module path.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .._lib import raw_stream as _raw_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _fused_sync_ok(bn, sums):
    """The two-launch statistics exchange (sync_bn.combine_forward_sums): fp32 affine parameters / running statistics on the
    device of the sums (PCM_SYNC_BN_FUSED=0: the framework-op route, for A/B)."""
    import os

    return (os.environ.get("PCM_SYNC_BN_FUSED", "1") != "0" and sums.is_cuda and bn.weight.dtype == torch.float32
            and bn.bias.dtype == torch.float32 and bn.weight.device == sums.device
            and (bn.running_mean is None or bn.running_mean.dtype == torch.float32))


class _BNReLU(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, sync_bn, relu=True):
        L = _lib.load()
        n, c = y.shape
        y = y.contiguous()
        dev = y.device
        st = _raw_stream()
        count = None
        ctx.relu = bool(relu)
        if n == 0:
            # a rank without rows (ragged data-parallel batches): nothing to launch, but with synchronised statistics the rank
            # must still take part in the collectives -- with count 0, like torch's SyncBatchNorm -- or its peers block forever
            assert sync_bn is not None, "bn_relu on an empty batch is only meaningful with synchronised statistics"
            from . import sync_bn as S

            zero = torch.zeros(2, c, dtype=torch.float32, device=dev)
            if _fused_sync_ok(sync_bn, zero):
                # the SAME collective API and the SAME combine kernel as the ranks that have rows (count 0: the pack kernel emits
                # zeros and never reads sums / src): every rank issues all_gather_into_tensor and updates its running
                # statistics with identical fp32 arithmetic, so the replicas' buffers stay bit-identical
                with torch.cuda.device(dev):
                    stat, count = S.combine_forward_sums(sync_bn, zero, zero, 0)
            else:
                stat, count = S.combine_forward(sync_bn, zero[0], zero[1], 0)
            ctx.save_for_backward(y, stat)
            ctx.partial, ctx.sync = None, (sync_bn, count)
            return torch.empty_like(y)
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            partial = torch.empty(L.pcm_bn_relu_slots(n, c) * 2 * c, **f32)
            sums, stat = torch.empty(2, c, **f32), torch.empty(4, c, **f32)
            z = torch.empty_like(y)
            args = (n, c, int(y.dtype == torch.bfloat16), int(bool(relu)), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum))
            if sync_bn is None:
                rc = L.pcm_bn_act_forward_hip(*args, _ptr(running_mean), _ptr(running_var), 0, partial.data_ptr(), sums.data_ptr(),
                                               stat.data_ptr(), z.data_ptr(), st)
            else:  # synchronised BatchNorm (policy/sync_bn.py): local sums -> statistics of all ranks -> apply
                from . import sync_bn as S

                rc = L.pcm_bn_act_forward_hip(*args, 0, 0, 2, partial.data_ptr(), sums.data_ptr(), 0, 0, st)
                _lib.check(rc, "pcm_bn_relu_forward_hip")
                if _fused_sync_ok(sync_bn, sums):
                    stat, count = S.combine_forward_sums(sync_bn, sums, y, n)  # the kernel accumulates around the first row of y
                else:
                    shift = y[0].float()
                    d = sums[0] / n
                    stat, count = S.combine_forward(sync_bn, shift + d, sums[1] - sums[0] * d, n)
                rc = L.pcm_bn_act_forward_hip(*args, 0, 0, 1, 0, 0, stat.data_ptr(), z.data_ptr(), st)
        _lib.check(rc, "pcm_bn_relu_forward_hip")
        ctx.save_for_backward(y, stat)
        ctx.partial = partial
        ctx.sync = (sync_bn, count)
        return z

    @staticmethod
    def backward(ctx, dz):
        L = _lib.load()
        y, stat = ctx.saved_tensors
        n, c = y.shape
        if n == 0:  # see forward: contribute zero sums to the exchange, no local gradient
            from . import sync_bn as S

            sync_bn, count = ctx.sync
            sums = torch.zeros(2, c, dtype=torch.float32, device=y.device)
            S.reduce_backward(sync_bn, sums, count)
            return torch.empty_like(y), sums[1], sums[0], None, None, None, None, None, None
        dz = dz.contiguous()
        if dz.dtype != y.dtype:
            dz = dz.to(y.dtype)
        dev = y.device
        with torch.cuda.device(dev):
            sums = torch.empty(2, c, dtype=torch.float32, device=dev)
            dy = torch.empty_like(y)
            st = _raw_stream()
            args = (n, c, int(y.dtype == torch.bfloat16), int(ctx.relu), y.data_ptr(), dz.data_ptr(), stat.data_ptr(), ctx.partial.data_ptr())
            sync_bn, count = ctx.sync
            if sync_bn is None:
                rc = L.pcm_bn_act_backward_hip(*args, sums.data_ptr(), dy.data_ptr(), 0, 0.0, st)
            else:  # local sums stay the parameter gradients; the input gradient uses the sums / count of all ranks
                from . import sync_bn as S

                rc = L.pcm_bn_act_backward_hip(*args, sums.data_ptr(), 0, 1, 0.0, st)
                _lib.check(rc, "pcm_bn_relu_backward_hip")
                gsums = S.reduce_backward(sync_bn, sums, count)  # count = n_loc / N: the kernel's 1 / n_loc becomes 1 / N
                rc = L.pcm_bn_act_backward_hip(*args, gsums.data_ptr(), dy.data_ptr(), 2, 0.0, st)
        _lib.check(rc, "pcm_bn_relu_backward_hip")
        return dy, sums[1], sums[0], None, None, None, None, None, None


def supported_layer(bn, C):
    """The part of ``supported`` that depends on the MODULE and the channel count alone (what a caller can check before any layer has run)."""
    return (type(bn) is torch.nn.BatchNorm1d and bn.affine and bn.track_running_stats and bn.weight.dtype == torch.float32
            and (not bn.training or bn.momentum is not None) and int(C) == bn.num_features
            and bool(_lib.load().pcm_bn_relu_supported(1, int(C))))


def supported(y, bn):
    if y.is_cuda and y.dim() == 2 and y.shape[0] == 0 and type(bn) is torch.nn.BatchNorm1d and bn.training:
        from .sync_bn import wants_sync

        return wants_sync(bn)  # an empty rank of a synchronised BatchNorm still joins the collectives (bn_relu handles it)
    return (y.is_cuda and y.dim() == 2 and y.dtype in (torch.float32, torch.bfloat16) and type(bn) is torch.nn.BatchNorm1d
            and bn.affine and bn.track_running_stats and bn.weight.dtype == torch.float32
            and (not bn.training or bn.momentum is not None)
            and bool(_lib.load().pcm_bn_relu_supported(int(y.shape[0]), int(y.shape[1]))))


def bn_relu(y, bn, relu=True):
    """relu(bn(y)) -- or bn(y) alone with relu=False (the last layer of the Diffusion Policy's projector) --; the caller checked
    ``supported(y, bn)``."""
    if bn.training:
        from .sync_bn import wants_sync

        if y.shape[0] == 1 and not wants_sync(bn):  # what torch.nn.functional.batch_norm (and so the reference) raises for one row
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(y.shape),))
        z = _BNReLU.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, bn if wants_sync(bn) else None, relu)
        from . import fused_ops

        fused_ops.count_batch(bn)
        return z
    # eval: the affine comes from the running statistics; same apply kernel, differentiable through framework ops
    if torch.is_grad_enabled() and (y.requires_grad or bn.weight.requires_grad):
        return torch.relu(bn(y)) if relu else bn(y)
    L = _lib.load()
    n, c = y.shape
    y = y.contiguous()
    with torch.cuda.device(y.device), torch.no_grad():
        invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        a = bn.weight.float() * invstd
        stat = torch.stack([bn.running_mean.float(), invstd, a, bn.bias.float() - a * bn.running_mean.float()]).contiguous()
        z = torch.empty_like(y)
        rc = L.pcm_bn_act_forward_hip(n, c, int(y.dtype == torch.bfloat16), int(bool(relu)), y.data_ptr(), bn.weight.data_ptr(),
                                      bn.bias.data_ptr(), float(bn.eps), 0.0, 0, 0, 1, 0, 0, stat.data_ptr(), z.data_ptr(), _raw_stream())
    _lib.check(rc, "pcm_bn_relu_forward_hip")
    return z
