"""subtraction -- mirrors /root/reference/libs/pointops/functions/subtraction.py:6-37."""
import torch
from torch.autograd import Function

from . import _common as C


class _Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        assert input1.is_contiguous() and input2.is_contiguous()
        C.require_hip(input1, input2, idx)
        C.f32c(input1, "input1")
        C.f32c(input2, "input2")
        L = C.lib()
        n, c = input1.shape
        nsample = idx.shape[-1]
        idx = C.i32c(idx)
        with torch.cuda.device(input1.device):
            output = torch.empty(n, nsample, c, dtype=torch.float32, device=input1.device)
            rc = L.pcm_subtraction_forward_hip(n, nsample, c, C.ptr(input1), C.ptr(input2), C.ptr(idx), C.ptr(output), C.stream())
        C._lib.check(rc, "pcm_subtraction_forward_hip")
        ctx.n2 = input2.shape[0]
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        n, nsample, c = grad_output.shape
        with torch.cuda.device(grad_output.device):
            g1 = torch.empty(n, c, dtype=torch.float32, device=grad_output.device)
            g2 = torch.empty(ctx.n2, c, dtype=torch.float32, device=grad_output.device)
            rc = L.pcm_subtraction_backward_hip(n, nsample, c, C.ptr(idx), C.ptr(grad_output), C.ptr(g1), 0, C.stream())
            C._lib.check(rc, "pcm_subtraction_backward_hip")
            C.segment_sum(g2, grad_output.view(n * nsample, c), plan=C.ScatterPlan(idx, ctx.n2), sign=-1.0)
        return g1, g2, None


subtraction = _Subtraction.apply
