"""grouping / grouping2 -- mirrors /root/reference/libs/pointops/functions/grouping.py:6-62."""
import torch
from torch.autograd import Function

from . import _common as C


class _Grouping2(Function):
    """grouping.py:6-32 (Grouping): plain gather (no -1 handling); the backward scatter is a planned segmented sum
    (csrc/segsum.hip) instead of the reference's one atomicAdd per element."""

    @staticmethod
    def forward(ctx, input, idx):
        assert input.is_contiguous() and idx.is_contiguous()
        C.require_hip(input, idx)
        C.f32c(input, "input")
        L = C.lib()
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        with torch.cuda.device(input.device):
            output = torch.empty(m, nsample, c, dtype=torch.float32, device=input.device)
            rc = L.pcm_grouping_forward_hip(m, nsample, c, C.ptr(input), C.ptr(idx), C.ptr(output), C.stream())
        C._lib.check(rc, "pcm_grouping_forward_hip")
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        m, nsample, c = grad_output.shape
        with torch.cuda.device(grad_output.device):
            grad_input = torch.empty(ctx.n, c, dtype=torch.float32, device=grad_output.device)
            C.segment_sum(grad_input, grad_output, plan=C.ScatterPlan(idx, ctx.n))
        return grad_input, None


grouping2 = _Grouping2.apply


class _GroupXYZFeat(Function):
    """Fused form of the pure-PyTorch grouping() (grouping.py:35-59): one kernel writes
    [ (xyz[idx] - new_xyz) * (idx != -1), feat[idx] or 0 ] instead of 2 cats + 2 gathers + einsum + cat."""

    @staticmethod
    def forward(ctx, feat, xyz, new_xyz, idx, with_xyz):
        L = C.lib()
        m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
        w = c + (3 if with_xyz else 0)
        with torch.cuda.device(feat.device):
            out = torch.empty(m, nsample, w, dtype=torch.float32, device=feat.device)
            rc = L.pcm_group_xyz_feat_forward_hip(
                m, nsample, c, C.ptr(xyz) if with_xyz else 0, C.ptr(new_xyz) if with_xyz else 0, C.ptr(feat),
                C.ptr(idx), C.ptr(out), C.stream(),
            )
        C._lib.check(rc, "pcm_group_xyz_feat_forward_hip")
        ctx.with_xyz = with_xyz
        ctx.n = feat.shape[0]
        ctx.n_xyz = xyz.shape[0]
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        L = C.lib()
        grad_out = grad_out.contiguous()
        m, nsample, w = grad_out.shape
        c = w - (3 if ctx.with_xyz else 0)
        grad_feat = grad_xyz = grad_new_xyz = None
        with torch.cuda.device(grad_out.device):
            if ctx.needs_input_grad[0]:
                # rows with idx == -1 are skipped by the plan; every feature row is written exactly once
                grad_feat = torch.empty(ctx.n, c, dtype=torch.float32, device=grad_out.device)
                C.segment_sum(grad_feat, grad_out, src_stride=w, src_off=w - c, plan=C.ScatterPlan(idx, ctx.n))
            if ctx.with_xyz and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                # coordinates rarely need gradients (never on the BC path); plain torch ops
                valid = (idx >= 0).to(grad_out.dtype).unsqueeze(-1)
                g_rel = grad_out[..., :3] * valid
                if ctx.needs_input_grad[2]:
                    grad_new_xyz = -g_rel.sum(dim=1)
                if ctx.needs_input_grad[1]:
                    grad_xyz = torch.zeros(ctx.n_xyz, 3, dtype=grad_out.dtype, device=grad_out.device)
                    grad_xyz.index_add_(0, idx.clamp_min(0).reshape(-1).long(), g_rel.reshape(-1, 3))
        return grad_feat, grad_xyz, grad_new_xyz, None, None


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """Same signature and result as grouping.py:35-59: idx == -1 selects an all-zero row, relative
    coordinates are masked for those rows, xyz channels come first."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    C.require_hip(idx, feat, xyz, new_xyz)
    C.f32c(feat, "feat")
    C.f32c(xyz, "xyz")
    if with_xyz:
        assert new_xyz.is_contiguous()
        C.f32c(new_xyz, "new_xyz")
    idx32 = C.i32c(idx)
    return _GroupXYZFeat.apply(feat, xyz, new_xyz, idx32, bool(with_xyz))
