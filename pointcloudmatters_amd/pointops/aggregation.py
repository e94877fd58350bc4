"""aggregation -- mirrors /root/reference/libs/pointops/functions/aggregation.py:6-56."""
import torch
from torch.autograd import Function

from . import _common as C


class _Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        C.require_hip(input, position, weight, idx)
        L = C.lib()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        idx = C.i32c(idx)
        with torch.cuda.device(input.device):
            output = torch.zeros(n, c, dtype=torch.float32, device=input.device)
            rc = L.pcm_aggregation_forward_hip(
                n, nsample, c, w_c, C.ptr(input), C.ptr(position), C.ptr(weight), C.ptr(idx), C.ptr(output), C.stream()
            )
        C._lib.check(rc, "pcm_aggregation_forward_hip")
        ctx.save_for_backward(input, position, weight, idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        dev = grad_output.device
        with torch.cuda.device(dev):
            gi = torch.empty(input.shape[0], c, dtype=torch.float32, device=dev)
            gp = torch.empty(n, nsample, c, dtype=torch.float32, device=dev)
            gw = torch.zeros(n, nsample, w_c, dtype=torch.float32, device=dev)
            rc = L.pcm_aggregation_backward_hip(
                n, nsample, c, w_c, C.ptr(input), C.ptr(position), C.ptr(weight), C.ptr(idx), C.ptr(grad_output),
                0, C.ptr(gp), C.ptr(gw), C.stream(),
            )  # grad_position and grad_weight: per-row, no scatter
            C._lib.check(rc, "pcm_aggregation_backward_hip")
            # grad_input[j, c] = sum over the (n, s) pairs with idx[n, s] == j of grad_output[n, c] * weight[n, s, c % w_c]
            C.segment_sum(gi, grad_output, plan=C.ScatterPlan(idx, input.shape[0]), rowdiv=nsample, scale=weight, scale_mode=2, w_c=w_c)
        return gi, gp, gw, None


aggregation = _Aggregation.apply
