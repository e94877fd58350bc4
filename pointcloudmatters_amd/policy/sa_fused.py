"""Fused set-abstraction layer (csrc/sa_fused.hip) as an autograd function.

    tokens = max_s relu( BN( Linear([p_j - q_i, f_j]) ) )        j = knn_idx[i, s]

is evaluated as  Gf = f @ Wf^T  (one GEMM on the n points, hipBLASLt; bf16 under autocast) followed by
ONE gather pass that adds the fp32 xyz term Wp (p_j - q_i), accumulates the BatchNorm batch
statistics and keeps per-(query, channel) max / min / arg -- see the kernel file for the algebra.
Same result as ``sa_impl="reference"`` up to fp32 re-association (tests: 1e-4 relative), ~K times
less HBM traffic and no (m, K, 3+C) / (m, H, K) intermediates.
"""
import os

import torch
from torch.autograd import Function

from .. import _lib
from .rows_linear import linear_rows
from .._lib import raw_stream as _raw_stream

# How the index statistics and the m*H backward deltas are accumulated:
#   "sorted"  (default) csrc/sa_scatter.hip: a CSR of the neighbour lists with sorted segments, sums taken in list order --
#             no float atomics anywhere, two runs of the same step give bit-identical gradients;
#   "atomic"  csrc/sa_fused.hip's LDS / global float atomics (the reference's own backward is atomic too,
#             libs/pointops/src/grouping/grouping_cuda_kernel.cu:24): a little less traffic, last bits vary from run to run.
SCATTER_MODE = os.environ.get("PCM_SA_SCATTER", "sorted")


def set_scatter_mode(mode):
    global SCATTER_MODE
    assert mode in ("sorted", "atomic"), mode
    old, SCATTER_MODE = SCATTER_MODE, mode
    return old


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def index_stats(p, q, knn_idx, o32=None, no32=None, n_max=0):
    """cnt (n), S (n,3), RM (12) of the neighbour lists -- the index-only part of the backward (occurrence count and summed
    relative coordinates of every point, global moments).  A function of the coordinates alone: sample_and_query runs it
    right behind the kNN query on the side stream, off the critical path.  Returns (ent, stats) in "atomic" mode and
    (ent, stats, csr) in "sorted" mode, csr = start (n+1) | list (m*K) of the sorted inverse neighbour lists."""
    L = _lib.load()
    n = p.shape[0]
    m, K = knn_idx.shape
    b = int(o32.shape[0]) if o32 is not None else 0
    if SCATTER_MODE == "sorted" and L.pcm_sa_det_supported(K, 4):
        with torch.cuda.device(p.device):
            buf = torch.empty(4 * n + 12, dtype=torch.float32, device=p.device)
            ent = torch.empty(m, K, 4, dtype=torch.float32, device=p.device)
            csr = torch.empty(n + 1 + m * K, dtype=torch.int32, device=p.device)
            scratch = torch.empty(L.pcm_sa_index_det_scratch_ints(n), dtype=torch.int32, device=p.device)
            cnt, S, RM = buf[:n], buf[n: 4 * n], buf[4 * n:]
            rc = L.pcm_sa_index_det_hip(m, K, n, p.data_ptr(), q.data_ptr(), knn_idx.data_ptr(), ent.data_ptr(), csr.data_ptr(),
                                        scratch.data_ptr(), cnt.data_ptr(), S.data_ptr(), RM.data_ptr(),
                                        _raw_stream())
        _lib.check(rc, "pcm_sa_index_det_hip")
        return ent, buf, csr
    with torch.cuda.device(p.device):
        buf = torch.zeros(4 * n + 12, dtype=torch.float32, device=p.device)
        ent = torch.empty(m, K, 4, dtype=torch.float32, device=p.device)  # (j bits, rel x, rel y, rel z) per neighbour slot
        cnt, S, RM = buf[:n], buf[n: 4 * n], buf[4 * n:]
        rc = L.pcm_sa_index_hip(m, K, p.data_ptr(), q.data_ptr(), knn_idx.data_ptr(), _ptr(o32), _ptr(no32), b, int(n_max),
                                ent.data_ptr(), cnt.data_ptr(), S.data_ptr(), RM.data_ptr(), _raw_stream())
    _lib.check(rc, "pcm_sa_index_hip")
    return ent, buf


def _slots(L, m, n, H, bf, K, b):
    return max(L.pcm_sa_fused_slots(m, H, bf, K), L.pcm_sa_fused_slots(m, H, 0, 1), L.pcm_sa_fused_slots(n, H, bf, 1),
               L.pcm_sa_bwd1_det_slots(m), b, 1) + L.pcm_sa_fused_reduce_scratch_rows()  # + second-level rows of the reductions


class _SAFused(Function):
    @staticmethod
    def forward(ctx, gf, ent, wp, gamma, beta, running_mean, running_var, eps, momentum, o32, no32, n_max, istats, sync_bn, csr=None):
        L = _lib.load()
        assert gf.is_cuda and gf.is_contiguous() and gf.dtype in (torch.float32, torch.bfloat16)
        n, H = gf.shape
        m, K = ent.shape[:2]
        dev = gf.device
        bf = 1 if gf.dtype == torch.bfloat16 else 0
        slots = _slots(L, m, n, H, bf, K, int(o32.shape[0]) if o32 is not None else 0)
        st = _raw_stream()
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            sel = torch.empty(m, H, **f32)
            asel = torch.empty(m, H, dtype=torch.uint8, device=dev)
            partial = torch.empty(slots * 5 * H, **f32)
            sums, stat, z = torch.empty(2, H, **f32), torch.empty(4, H, **f32), torch.empty(m, H, **f32)
            wp = wp.contiguous().float()
            gamma, beta = gamma.contiguous().float(), beta.contiguous().float()
            count = None

            def launch(mask, stat_t, rm=None, rv=None):
                rc_ = L.pcm_sa_fused_forward_hip(
                    m, K, H, bf, gf.data_ptr(), ent.data_ptr(),
                    wp.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum), _ptr(rm),
                    _ptr(rv), sel.data_ptr(), asel.data_ptr(), partial.data_ptr(),
                    sums.data_ptr(), stat_t.data_ptr(), z.data_ptr(), mask, st)
                _lib.check(rc_, "pcm_sa_fused_forward_hip")

            if sync_bn is None:
                launch(0, stat, running_mean, running_var)
            else:  # synchronised BatchNorm: local sums (around the first neighbour's Gf row) -> all ranks -> apply
                from . import sync_bn as S

                launch(1 | 2, stat)
                rows = float(m) * K
                from .bn_relu import _fused_sync_ok

                if _fused_sync_ok(sync_bn, sums) and gf.dtype in (torch.float32, torch.bfloat16):
                    # the kernel accumulated around the first neighbour's Gf row: its index is the first word of `ent`
                    stat, count = S.combine_forward_sums(sync_bn, sums, gf, rows, row_index=ent.view(torch.int32).reshape(-1)[:1])
                else:
                    j0 = ent.view(torch.int32)[0, 0, 0]
                    shift = torch.where(j0 >= 0, gf.index_select(0, j0.clamp(min=0).long().reshape(1))[0].float(),
                                        torch.zeros(H, **f32))
                    d = sums[0] / rows
                    stat, count = S.combine_forward(sync_bn, shift + d, sums[1] - sums[0] * d, rows)
                launch(8, stat)
        ctx.save_for_backward(gf, ent, wp, stat, sel, asel, istats)
        ctx.csr = csr if (csr is not None and L.pcm_sa_det_supported(K, H)) else None
        ctx.partial = partial
        ctx.layout = (o32, no32, int(n_max))
        ctx.sync = (sync_bn, count)
        ctx.mark_non_differentiable(stat)
        return z, stat

    @staticmethod
    def backward(ctx, dz, _dstat):
        L = _lib.load()
        gf, ent, wp, stat, sel, asel, istats = ctx.saved_tensors
        n, H = gf.shape
        m, K = ent.shape[:2]
        dev = gf.device
        st = _raw_stream()
        dz = dz.contiguous().float()
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            o32, no32, n_max = ctx.layout
            b = int(o32.shape[0]) if o32 is not None else 0
            csr = ctx.csr
            lds_path = csr is None and b > 0 and L.pcm_sa_fused_bwd1_lds_channels(H, n_max) > 0
            # the sorted and the LDS-staged scatter write every element of D themselves; the global-atomic fallback adds into zeros
            D = torch.empty(n * H, **f32) if (lds_path or csr is not None) else torch.zeros(n * H, **f32)
            cnt, S, RM = istats[:n], istats[n: 4 * n], istats[4 * n:]
            red1, red2 = torch.empty(5, H, **f32), torch.empty(3, H, **f32)
            dgf = torch.empty_like(gf)
            dwp, dgamma, dbeta = torch.empty(H, 3, **f32), torch.empty(H, **f32), torch.empty(H, **f32)
            sync_bn, count = ctx.sync

            def launch(mask, red_g=None):
                if csr is not None and (mask <= 0 or mask & 6):  # stages 2|4 (delta scatter + its sums) without atomics
                    ws = torch.empty(L.pcm_sa_bwd1_det_ws_bytes(m, K, H), dtype=torch.uint8, device=dev)
                    rc_ = L.pcm_sa_bwd1_det_hip(m, n, K, H, dz.data_ptr(), sel.data_ptr(), asel.data_ptr(), stat.data_ptr(),
                                                ent.data_ptr(), csr.data_ptr(), ws.data_ptr(), D.data_ptr(), ctx.partial.data_ptr(),
                                                red1.data_ptr(), 0, st)
                    _lib.check(rc_, "pcm_sa_bwd1_det_hip")
                    mask = (0x3E if mask <= 0 else mask) & ~6
                    if not mask:
                        return
                rc_ = L.pcm_sa_fused_backward_hip(
                    m, n, K, H, 1 if gf.dtype == torch.bfloat16 else 0, gf.data_ptr(), ent.data_ptr(),
                    wp.data_ptr(), stat.data_ptr(), dz.data_ptr(), sel.data_ptr(), asel.data_ptr(), D.data_ptr(), cnt.data_ptr(),
                    S.data_ptr(), RM.data_ptr(), ctx.partial.data_ptr(), red1.data_ptr(), red2.data_ptr(), dgf.data_ptr(),
                    dwp.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(o32) if lds_path else 0,
                    _ptr(no32) if lds_path else 0, b if lds_path else 0, n_max, _ptr(red_g), 0.0, mask, st)
                _lib.check(rc_, "pcm_sa_fused_backward_hip")

            if sync_bn is None:
                launch(0)
            else:  # {sum delta, sum delta * yhat} of all ranks enter the input gradient; dgamma / dbeta stay local
                from . import sync_bn as SB

                launch(2 | 4)
                red_g = SB.reduce_backward(sync_bn, red1[:2], count)  # count = rows_loc / N_global (device scalar)
                launch(8 | 16 | 32, red_g)
        return dgf, None, dwp, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def supports(owner, x):
    bn = owner.bn
    if not (x.is_cuda and type(bn) is torch.nn.BatchNorm1d and bn.track_running_stats and bn.affine):
        return False
    if owner.training:
        return bn.momentum is not None
    # eval (rollout): running statistics, forward only
    return not (torch.is_grad_enabled() and (x.requires_grad or owner.linear.weight.requires_grad))


def _sa_fused_eval(owner, gf, ent, wp):
    """Inference form: the BatchNorm affine comes from the running statistics, so the batch-statistics stages of
    the forward launcher are skipped (stage_mask = gather | apply) and nothing is saved for backward."""
    L = _lib.load()
    bn = owner.bn
    n, H = gf.shape
    m, K = ent.shape[:2]
    dev = gf.device
    bf = 1 if gf.dtype == torch.bfloat16 else 0
    slots = L.pcm_sa_fused_slots(m, H, bf, K) + L.pcm_sa_fused_reduce_scratch_rows()
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        a = bn.weight.float() * invstd
        stat = torch.stack([bn.running_mean.float(), invstd, a, bn.bias.float() - a * bn.running_mean.float()]).contiguous()
        sel = torch.empty(m, H, **f32)
        asel = torch.empty(m, H, dtype=torch.uint8, device=dev)
        partial = torch.empty(slots * 5 * H, **f32)
        z = torch.empty(m, H, **f32)
        wp = wp.contiguous().float()
        gamma = bn.weight.contiguous().float()
        rc = L.pcm_sa_fused_forward_hip(
            m, K, H, bf, gf.data_ptr(), ent.data_ptr(),
            wp.data_ptr(), gamma.data_ptr(), bn.bias.data_ptr(), float(bn.eps), 0.0, 0, 0, sel.data_ptr(),
            asel.data_ptr(), partial.data_ptr(), 0, stat.data_ptr(), z.data_ptr(), 1 | 8,
            _raw_stream())
    _lib.check(rc, "pcm_sa_fused_forward_hip")
    return z


def layout_of(o, n_o):
    """(offset int32, new_offset int32, largest cloud) of a packed batch, or (None, None, 0)."""
    from ..pointops import _common as C

    if o is None or n_o is None:
        return None, None, 0
    return C.i32c(o), C.i32c(n_o), max(C.counts_from_offsets(C.host_offsets(o)))


def sa_fused_forward(owner, p, x, n_p, fps_idx, knn_idx, o=None, n_o=None, istats=None):
    """tokens (m, H) of the SA layer owned by `owner` (linear, bn) for features x (n, C).  `o` / `n_o`
    (cumulative offsets of points / queries per cloud) enable the LDS-staged backward scatter; `istats` = the
    index_stats() buffer of these neighbour lists when it was computed ahead of time."""
    o32, no32, n_max = layout_of(o, n_o)
    w = owner.linear.weight  # (H, 3 + C): xyz columns first (grouping.py:57 concatenates xyz before feat)
    gf = linear_rows(x, w[:, 3:])  # (n, H); bf16 under autocast, fp32 otherwise
    if gf.dtype not in (torch.float32, torch.bfloat16):
        gf = gf.float()
    bn = owner.bn
    if istats is None:
        with torch.no_grad():
            istats = index_stats(p, n_p, knn_idx, o32, no32, n_max)
    ent, stats = istats[0], istats[1]
    csr = istats[2] if len(istats) > 2 else None
    if not owner.training:
        with torch.no_grad():
            return _sa_fused_eval(owner, gf.contiguous(), ent, w[:, :3])
    from .sync_bn import wants_sync

    z, _ = _SAFused.apply(gf.contiguous(), ent, w[:, :3], bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.eps, bn.momentum, o32, no32, n_max, stats, bn if wants_sync(bn) else None, csr)
    from . import fused_ops

    fused_ops.count_batch(bn)
    return z
