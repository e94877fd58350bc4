"""``pointops._C`` for MI355X: the 16 entry points of the reference's pybind module
(/root/reference/libs/pointops/src/pointops_api.cpp:15-32) with their exact positional signatures
(``int dims..., [float radii...], Tensor in..., Tensor out...`` -> None), implemented by passing
``tensor.data_ptr()`` and the current HIP stream to the C ABI of include/pcm_pointops.h.

With this module in place the reference's own wrappers (libs/pointops/functions/*.py: allocation with
``torch.cuda.IntTensor``, ``tmp.fill_(1e10)``, ``offset.int()`` ...) run unchanged on PyTorch-ROCm.
Like the reference: caller-owned, pre-allocated / pre-filled buffers, no validation beyond dtype.
"""
import torch

from .. import _lib
from .._lib import raw_stream as _raw_stream


def _st():
    return _raw_stream()


def _f(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected a float32 tensor, got {t.dtype}")  # data_ptr<float>() throws in the reference
    return t.data_ptr()


def _i(t):
    if t.dtype != torch.int32:
        raise TypeError(f"expected an int32 tensor, got {t.dtype}")
    return t.data_ptr()


def _run(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args, _st()), name)


def farthest_point_sampling_cuda(b, n, xyz, offset, new_offset, tmp, idx):
    _run("pcm_farthest_point_sampling_hip", int(b), int(n), _f(xyz), _i(offset), _i(new_offset), _f(tmp), _i(idx))


def knn_query_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    _run("pcm_knn_query_hip", int(m), int(nsample), _f(xyz), _f(new_xyz), _i(offset), _i(new_offset), _i(idx), _f(dist2))


def ball_query_cuda(m, nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset, idx, dist2):
    _run("pcm_ball_query_hip", int(m), int(nsample), float(min_radius), float(max_radius), _f(xyz), _f(new_xyz),
         _i(offset), _i(new_offset), _i(idx), _f(dist2))


def random_ball_query_cuda(m, nsample, min_radius, max_radius, order, xyz, new_xyz, offset, new_offset, idx, dist2):
    _run("pcm_random_ball_query_hip", int(m), int(nsample), float(min_radius), float(max_radius), _i(order), _f(xyz),
         _f(new_xyz), _i(offset), _i(new_offset), _i(idx), _f(dist2))


def grouping_forward_cuda(m, nsample, c, input, idx, output):
    _run("pcm_grouping_forward_hip", int(m), int(nsample), int(c), _f(input), _i(idx), _f(output))


def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
    _run("pcm_grouping_backward_hip", int(m), int(nsample), int(c), _f(grad_output), _i(idx), _f(grad_input))


def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
    _run("pcm_interpolation_forward_hip", int(n), int(c), int(k), _f(input), _i(idx), _f(weight), _f(output))


def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
    _run("pcm_interpolation_backward_hip", int(n), int(c), int(k), _f(grad_output), _i(idx), _f(weight), _f(grad_input))


def subtraction_forward_cuda(n, nsample, c, input1, input2, idx, output):
    _run("pcm_subtraction_forward_hip", int(n), int(nsample), int(c), _f(input1), _f(input2), _i(idx), _f(output))


def subtraction_backward_cuda(n, nsample, c, idx, grad_output, grad_input1, grad_input2):
    _run("pcm_subtraction_backward_hip", int(n), int(nsample), int(c), _i(idx), _f(grad_output), _f(grad_input1), _f(grad_input2))


def aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, output):
    _run("pcm_aggregation_forward_hip", int(n), int(nsample), int(c), int(w_c), _f(input), _f(position), _f(weight), _i(idx), _f(output))


def aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight):
    _run("pcm_aggregation_backward_hip", int(n), int(nsample), int(c), int(w_c), _f(input), _f(position), _f(weight), _i(idx),
         _f(grad_output), _f(grad_input), _f(grad_position), _f(grad_weight))


def attention_relation_step_forward_cuda(m, g, c, query, key, weight, index_target, index_refer, output):
    _run("pcm_attention_relation_step_forward_hip", int(m), int(g), int(c), _f(query), _f(key), _f(weight), _i(index_target),
         _i(index_refer), _f(output))


def attention_relation_step_backward_cuda(m, g, c, query, grad_query, key, grad_key, weight, grad_weight, index_target,
                                          index_refer, grad_output):
    _run("pcm_attention_relation_step_backward_hip", int(m), int(g), int(c), _f(query), _f(grad_query), _f(key), _f(grad_key),
         _f(weight), _f(grad_weight), _i(index_target), _i(index_refer), _f(grad_output))


def attention_fusion_step_forward_cuda(m, g, c, weight, value, index_target, index_refer, output):
    _run("pcm_attention_fusion_step_forward_hip", int(m), int(g), int(c), _f(weight), _f(value), _i(index_target),
         _i(index_refer), _f(output))


def attention_fusion_step_backward_cuda(m, g, c, weight, grad_weight, value, grad_value, index_target, index_refer, grad_output):
    _run("pcm_attention_fusion_step_backward_hip", int(m), int(g), int(c), _f(weight), _f(grad_weight), _f(value), _f(grad_value),
         _i(index_target), _i(index_refer), _f(grad_output))
