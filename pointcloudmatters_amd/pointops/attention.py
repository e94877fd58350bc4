"""attention_relation_step / attention_fusion_step -- mirrors
/root/reference/libs/pointops/functions/attention.py:11-119."""
import torch
from torch.autograd import Function

from . import _common as C


class _AttentionRelationStep(Function):
    @staticmethod
    def forward(ctx, query, key, weight, index_target, index_refer):
        assert query.is_contiguous() and key.is_contiguous() and weight.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        C.require_hip(query, key, weight, index_target, index_refer)
        L = C.lib()
        _, g, c = query.shape
        m = index_target.shape[0]
        it, ir = C.i32c(index_target), C.i32c(index_refer)
        with torch.cuda.device(query.device):
            output = torch.zeros(m, g, dtype=torch.float32, device=query.device)
            rc = L.pcm_attention_relation_step_forward_hip(
                m, g, c, C.ptr(query), C.ptr(key), C.ptr(weight), C.ptr(it), C.ptr(ir), C.ptr(output), C.stream()
            )
        C._lib.check(rc, "pcm_attention_relation_step_forward_hip")
        ctx.save_for_backward(query, key, weight, it, ir)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        query, key, weight, it, ir = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        n, g, c = query.shape
        m = it.shape[0]
        dev = query.device
        with torch.cuda.device(dev):
            gq = torch.zeros(n, g, c, dtype=torch.float32, device=dev)
            gk = torch.zeros(n, g, c, dtype=torch.float32, device=dev)
            gw = torch.zeros(c, dtype=torch.float32, device=dev)
            rc = L.pcm_attention_relation_step_backward_hip(
                m, g, c, C.ptr(query), C.ptr(gq), C.ptr(key), C.ptr(gk), C.ptr(weight), C.ptr(gw), C.ptr(it), C.ptr(ir),
                C.ptr(grad_output), C.stream(),
            )
        C._lib.check(rc, "pcm_attention_relation_step_backward_hip")
        return gq, gk, None, None, None  # attention.py:61: no gradient for weight


class _AttentionFusionStep(Function):
    @staticmethod
    def forward(ctx, weight, value, index_target, index_refer):
        assert weight.is_contiguous() and value.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        C.require_hip(weight, value, index_target, index_refer)
        L = C.lib()
        n, g, c = value.shape
        m = index_refer.shape[0]
        it, ir = C.i32c(index_target), C.i32c(index_refer)
        with torch.cuda.device(value.device):
            output = torch.zeros(n, g, c, dtype=torch.float32, device=value.device)
            rc = L.pcm_attention_fusion_step_forward_hip(
                m, g, c, C.ptr(weight), C.ptr(value), C.ptr(it), C.ptr(ir), C.ptr(output), C.stream()
            )
        C._lib.check(rc, "pcm_attention_fusion_step_forward_hip")
        ctx.save_for_backward(weight, value, it, ir)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        weight, value, it, ir = ctx.saved_tensors
        L = C.lib()
        grad_output = grad_output.contiguous()
        n, g, c = value.shape
        m = it.shape[0]
        dev = value.device
        with torch.cuda.device(dev):
            gw = torch.zeros(m, g, dtype=torch.float32, device=dev)
            gv = torch.zeros(n, g, c, dtype=torch.float32, device=dev)
            rc = L.pcm_attention_fusion_step_backward_hip(
                m, g, c, C.ptr(weight), C.ptr(gw), C.ptr(value), C.ptr(gv), C.ptr(it), C.ptr(ir), C.ptr(grad_output),
                C.stream(),
            )
        C._lib.check(rc, "pcm_attention_fusion_step_backward_hip")
        return gw, gv, None, None


attention_relation_step = _AttentionRelationStep.apply
attention_fusion_step = _AttentionFusionStep.apply
