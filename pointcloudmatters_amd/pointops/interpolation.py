"""interpolation / interpolation2 -- mirrors /root/reference/libs/pointops/functions/interpolation.py:8-59."""
import torch
from torch.autograd import Function

from . import _common as C
from .query import knn_query


def _weights(xyz, new_xyz, offset, new_offset, k):
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)  # (n, k), (n, k)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    return idx, (dist_recip / norm).contiguous()


class _Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        C.require_hip(xyz, new_xyz, input)
        C.f32c(input, "input")
        L = C.lib()
        idx, weight = _weights(xyz, new_xyz, offset, new_offset, k)
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        with torch.cuda.device(input.device):
            output = torch.empty(n, c, dtype=torch.float32, device=input.device)
            rc = L.pcm_interpolation_forward_hip(n, c, k, C.ptr(input), C.ptr(idx), C.ptr(weight), C.ptr(output), C.stream())
        C._lib.check(rc, "pcm_interpolation_forward_hip")
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        with torch.cuda.device(grad_output.device):
            # grad_input[j] = sum over the (n, i) pairs with idx[n, i] == j of grad_output[n] * weight[n, i]
            grad_input = torch.empty(ctx.m, c, dtype=torch.float32, device=grad_output.device)
            C.segment_sum(grad_input, grad_output, plan=C.ScatterPlan(idx, ctx.m), rowdiv=ctx.k, scale=weight, scale_mode=1)
        return None, None, grad_input, None, None, None


interpolation2 = _Interpolation.apply


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """interpolation.py:8-22 computes the same weighted sum with k torch gathers (autograd through
    feat); numerically identical to interpolation2 (accumulation in k order), so both share the kernel."""
    return _Interpolation.apply(xyz, new_xyz, feat, offset, new_offset, k)
