"""Fused set-abstraction layer (csrc/sa_fused.hip) as an autograd function.

    tokens = max_s relu( BN( Linear([p_j - q_i, f_j]) ) )        j = knn_idx[i, s]

is evaluated as  Gf = f @ Wf^T  (one GEMM on the n points, hipBLASLt; bf16 under autocast) followed by
ONE gather pass that adds the fp32 xyz term Wp (p_j - q_i), accumulates the BatchNorm batch
statistics and keeps per-(query, channel) max / min / arg -- see the kernel file for the algebra.
Same result as ``sa_impl="reference"`` up to fp32 re-association (tests: 1e-4 relative), ~K times
less HBM traffic and no (m, K, 3+C) / (m, H, K) intermediates.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .rows_linear import linear_rows


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _SAFused(Function):
    @staticmethod
    def forward(ctx, gf, p, q, knn_idx, wp, gamma, beta, running_mean, running_var, eps, momentum, o32, no32, n_max):
        L = _lib.load()
        assert gf.is_cuda and gf.is_contiguous() and gf.dtype in (torch.float32, torch.bfloat16)
        n, H = gf.shape
        m, K = knn_idx.shape
        dev = gf.device
        vec = 4 if H % 4 == 0 else 1
        slots = max(L.pcm_sa_fused_slots(m, H, vec), L.pcm_sa_fused_slots(n, H, vec), int(o32.shape[0]) if o32 is not None else 0)
        st = torch.cuda.current_stream().cuda_stream
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            ymax, ymin = torch.empty(m, H, **f32), torch.empty(m, H, **f32)
            amax = torch.empty(m, H, dtype=torch.uint8, device=dev)
            amin = torch.empty(m, H, dtype=torch.uint8, device=dev)
            partial = torch.empty(slots * 5 * H, **f32)
            sums, stat, z = torch.empty(2, H, **f32), torch.empty(4, H, **f32), torch.empty(m, H, **f32)
            wp = wp.contiguous().float()
            gamma, beta = gamma.contiguous().float(), beta.contiguous().float()
            rc = L.pcm_sa_fused_forward_hip(
                m, K, H, 1 if gf.dtype == torch.bfloat16 else 0, gf.data_ptr(), p.data_ptr(), q.data_ptr(), knn_idx.data_ptr(),
                wp.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum), _ptr(running_mean),
                _ptr(running_var), ymax.data_ptr(), ymin.data_ptr(), amax.data_ptr(), amin.data_ptr(), partial.data_ptr(),
                sums.data_ptr(), stat.data_ptr(), z.data_ptr(), 0, st)
        _lib.check(rc, "pcm_sa_fused_forward_hip")
        ctx.save_for_backward(gf, p, q, knn_idx, wp, stat, z, ymax, ymin, amax, amin)
        ctx.partial = partial
        ctx.layout = (o32, no32, int(n_max))
        ctx.mark_non_differentiable(stat)
        return z, stat

    @staticmethod
    def backward(ctx, dz, _dstat):
        L = _lib.load()
        gf, p, q, knn_idx, wp, stat, z, ymax, ymin, amax, amin = ctx.saved_tensors
        n, H = gf.shape
        m, K = knn_idx.shape
        dev = gf.device
        st = torch.cuda.current_stream().cuda_stream
        dz = dz.contiguous().float()
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            o32, no32, n_max = ctx.layout
            b = int(o32.shape[0]) if o32 is not None else 0
            lds_path = b > 0 and L.pcm_sa_fused_bwd1_lds_channels(H, n_max) > 0
            if lds_path:  # the LDS-staged scatter writes every element of D itself
                D = torch.empty(n * H, **f32)
                zeros = torch.zeros(4 * n + 12, **f32)
                cnt, S, RM = zeros[:n], zeros[n : 4 * n], zeros[4 * n :]
            else:
                zeros = torch.zeros(n * H + 4 * n + 12, **f32)  # D | cnt | S | RM in one memset
                D, cnt, S, RM = zeros[: n * H], zeros[n * H : n * H + n], zeros[n * H + n : n * H + 4 * n], zeros[n * H + 4 * n :]
            red1, red2 = torch.empty(5, H, **f32), torch.empty(3, H, **f32)
            dgf = torch.empty_like(gf)
            dwp, dgamma, dbeta = torch.empty(H, 3, **f32), torch.empty(H, **f32), torch.empty(H, **f32)
            rc = L.pcm_sa_fused_backward_hip(
                m, n, K, H, 1 if gf.dtype == torch.bfloat16 else 0, gf.data_ptr(), p.data_ptr(), q.data_ptr(), knn_idx.data_ptr(),
                wp.data_ptr(), stat.data_ptr(), dz.data_ptr(), z.data_ptr(), ymax.data_ptr(), ymin.data_ptr(), amax.data_ptr(),
                amin.data_ptr(), D.data_ptr(), cnt.data_ptr(), S.data_ptr(), RM.data_ptr(), ctx.partial.data_ptr(),
                red1.data_ptr(), red2.data_ptr(), dgf.data_ptr(), dwp.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                _ptr(o32) if lds_path else 0, _ptr(no32) if lds_path else 0, b if lds_path else 0, n_max, 0, st)
        _lib.check(rc, "pcm_sa_fused_backward_hip")
        return dgf, None, None, None, dwp, dgamma, dbeta, None, None, None, None, None, None, None


def supports(owner, x):
    bn = owner.bn
    if not (x.is_cuda and type(bn) is torch.nn.BatchNorm1d and bn.track_running_stats and bn.affine):
        return False
    if owner.training:
        return bn.momentum is not None
    # eval (rollout): running statistics, forward only
    return not (torch.is_grad_enabled() and (x.requires_grad or owner.linear.weight.requires_grad))


def _sa_fused_eval(owner, gf, p, n_p, knn_idx, wp):
    """Inference form: the BatchNorm affine comes from the running statistics, so the batch-statistics stages of
    the forward launcher are skipped (stage_mask = gather | apply) and nothing is saved for backward."""
    L = _lib.load()
    bn = owner.bn
    n, H = gf.shape
    m, K = knn_idx.shape
    dev = gf.device
    vec = 4 if H % 4 == 0 else 1
    slots = L.pcm_sa_fused_slots(m, H, vec)
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        a = bn.weight.float() * invstd
        stat = torch.stack([bn.running_mean.float(), invstd, a, bn.bias.float() - a * bn.running_mean.float()]).contiguous()
        ymax, ymin = torch.empty(m, H, **f32), torch.empty(m, H, **f32)
        amax = torch.empty(m, H, dtype=torch.uint8, device=dev)
        amin = torch.empty(m, H, dtype=torch.uint8, device=dev)
        partial = torch.empty(slots * 5 * H, **f32)
        z = torch.empty(m, H, **f32)
        wp = wp.contiguous().float()
        rc = L.pcm_sa_fused_forward_hip(
            m, K, H, 1 if gf.dtype == torch.bfloat16 else 0, gf.data_ptr(), p.data_ptr(), n_p.data_ptr(), knn_idx.data_ptr(),
            wp.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), 0.0, 0, 0, ymax.data_ptr(), ymin.data_ptr(),
            amax.data_ptr(), amin.data_ptr(), partial.data_ptr(), 0, stat.data_ptr(), z.data_ptr(), 1 | 8,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pcm_sa_fused_forward_hip")
    return z


def sa_fused_forward(owner, p, x, n_p, fps_idx, knn_idx, o=None, n_o=None):
    """tokens (m, H) of the SA layer owned by `owner` (linear, bn) for features x (n, C).  `o` / `n_o`
    (cumulative offsets of points / queries per cloud) enable the LDS-staged backward scatter."""
    from ..pointops import _common as C

    o32 = no32 = None
    n_max = 0
    if o is not None and n_o is not None:
        o32, no32 = C.i32c(o), C.i32c(n_o)
        n_max = max(C.counts_from_offsets(C.host_offsets(o)))
    w = owner.linear.weight  # (H, 3 + C): xyz columns first (grouping.py:57 concatenates xyz before feat)
    gf = linear_rows(x, w[:, 3:])  # (n, H); bf16 under autocast, fp32 otherwise
    if gf.dtype not in (torch.float32, torch.bfloat16):
        gf = gf.float()
    bn = owner.bn
    if not owner.training:
        with torch.no_grad():
            return _sa_fused_eval(owner, gf.contiguous(), p, n_p, knn_idx, w[:, :3])
    z, _ = _SAFused.apply(gf.contiguous(), p, n_p, knn_idx, w[:, :3], bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.eps, bn.momentum, o32, no32, n_max)
    with torch.no_grad():
        bn.num_batches_tracked.add_(1)
    return z
