"""ctypes binding of ``lib/libpcm_pointops.so`` -- the C-ABI library declared in
``include/pcm_pointops.h``.  There is NO fallback: if the library is missing or a symbol is absent
the import of any product op raises, and every op refuses non-HIP tensors.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpcm_pointops.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_P = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> argtypes, exactly the declarations of include/pcm_pointops.h (stream last, as void*)
SIGNATURES = {
    "pcm_opt_n_threads": [_i],
    "pcm_farthest_point_sampling_hip": [_i, _i, _P, _P, _P, _P, _P, _P],
    "pcm_knn_query_hip": [_i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_knn_query_b_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_knn_query_n_hip": [_i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_ball_query_hip": [_i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P],
    "pcm_random_ball_query_hip": [_i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_ball_query_b_hip": [_i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P],
    "pcm_ball_query_ws_bytes": [_i],
    "pcm_ball_query_ws_hip": [_i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "pcm_random_ball_query_b_hip": [_i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_grouping_forward_hip": [_i, _i, _i, _P, _P, _P, _P],
    "pcm_grouping_backward_hip": [_i, _i, _i, _P, _P, _P, _P],
    "pcm_interpolation_forward_hip": [_i, _i, _i, _P, _P, _P, _P, _P],
    "pcm_interpolation_backward_hip": [_i, _i, _i, _P, _P, _P, _P, _P],
    "pcm_subtraction_forward_hip": [_i, _i, _i, _P, _P, _P, _P, _P],
    "pcm_subtraction_backward_hip": [_i, _i, _i, _P, _P, _P, _P, _P],
    "pcm_aggregation_forward_hip": [_i, _i, _i, _i, _P, _P, _P, _P, _P, _P],
    "pcm_aggregation_backward_hip": [_i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_scatter_plan_ws_ints": [ctypes.c_long, _i],
    "pcm_scatter_plan_hip": [ctypes.c_long, _i, _P, _P, _P, _P, _P],
    "pcm_scatter_plan_sorted_scratch_ints": [_i],
    "pcm_scatter_plan_sorted_hip": [ctypes.c_long, _i, _P, _P, _P, _P, _P],
    "pcm_sa_det_supported": [_i, _i],
    "pcm_sa_index_det_scratch_ints": [_i],
    "pcm_sa_index_entries_hip": [_i, _i, _P, _P, _P, _P, _P],
    "pcm_sa_index_det_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_sa_bwd1_det_slots": [_i],
    "pcm_sa_bwd1_det_ws_bytes": [_i, _i, _i],
    "pcm_sa_bwd1_det_hip": [_i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P],
    "pcm_segment_sum_hip": [ctypes.c_long, _i, _P, _i, _P, _P, _i, _P, _i, _i, _f, _P, _i, _i, _P, _P],
    "pcm_attention_relation_step_forward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_attention_relation_step_backward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_attention_fusion_step_forward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P],
    "pcm_attention_fusion_step_backward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_group_xyz_feat_forward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P],
    "pcm_group_xyz_feat_backward_hip": [_i, _i, _i, _i, _P, _P, _P, _P],
    "pcm_sa_fused_slots": [_i, _i, _i, _i],
    "pcm_sa_fused_reduce_scratch_rows": [],
    "pcm_sa_reduce_rows_hip": [_i, _i, _P, _P, _P, _P],
    "pcm_sa_fused_bwd1_lds_channels": [_i, _i],
    "pcm_sa_fused_forward_hip": [_i, _i, _i, _i, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P],
    "pcm_sa_index_hip": [_i, _i, _P, _P, _P, _P, _P, _i, _i, _P, _P, _P, _P, _P],
    "pcm_sa_fused_backward_hip": [_i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                  _P, _P, _P, _i, _i, _P, ctypes.c_double, _i, _P],
    "pcm_bn_sync_pack_hip": [_i, ctypes.c_double, _P, _P, _i, _P, _P, _P],
    "pcm_bn_sync_combine_hip": [_i, _i, _P, _P, _P, _f, _f, _P, _P, ctypes.c_double, _P, _P, _P],
    "pcm_drln_blocks": [ctypes.c_long],
    "pcm_drln_forward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _f, _f, _P, ctypes.c_uint, _P, _P, _P, _P, _P],
    "pcm_drln_backward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _f, _P, ctypes.c_uint, _P, _P, _P, _P, _P, _P],
    "pcm_drln_forward2_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _f, _f, _P, ctypes.c_uint, _P, _P, _P, _P, _P, ctypes.c_long, _P, _P,
                              _P],
    "pcm_drln_backward2_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _f, _P, ctypes.c_uint, _P, _P, _P, _P, _P, _P],
    "pcm_proj_drln_mfma_supported": [_i, _i],
    # R, E, K, a, a_ls, W, bias, bias_is_bf16, x, gamma, beta, eps, p, seed, site, s, out, mean, rstd, pos, pos_n, sum16, out16, stream
    "pcm_proj_drln_mfma_forward_hip": [ctypes.c_long, _i, _i, _P, ctypes.c_long, _P, _P, _i, _P, _P, _P, _f, _f, _P, ctypes.c_uint, _P, _P,
                                       _P, _P, _P, ctypes.c_long, _P, _P, _P],
    "pcm_proj_drln_mfma_backward_supported": [_i, _i],
    "pcm_proj_drln_mfma_backward_blocks": [ctypes.c_long],
    # R, E, K, dout, dout2, s, mean, rstd, gamma, p, seed, site, W, dx, dy, da, da_ls, partial, dgamma_dbeta, dysum_bf16, stream
    "pcm_proj_drln_mfma_backward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _f, _P, ctypes.c_uint, _P, _P, _P, _P, ctypes.c_long, _P,
                                        _P, _P, _P],
    "pcm_linear_mfma_backward_supported": [_i, _i, _i],
    # R, N, K, dy, dy_ls, W, dres, dx, dpos, pos_cols, stream
    "pcm_linear_mfma_backward_hip": [ctypes.c_long, _i, _i, _P, ctypes.c_long, _P, _P, _P, _P, _i, _P],
    "pcm_linear_mfma_supported": [_i, _i, _i],
    # R, N, K, a, a_is_f32, a_ls, a_alt, pos, pos_n, pos_cols, W, bias, bias_is_bf16, out, out_is_bf16, out_ls, emit_pos16, emit_x16, stream
    "pcm_linear_mfma_forward_hip": [ctypes.c_long, _i, _i, _P, _i, ctypes.c_long, _P, _P, ctypes.c_long, _i, _P, _P, _i, _P, _i, ctypes.c_long,
                                    _P, _P, _P],
    "pcm_ffn_ln_supported": [_i, _i],
    "pcm_ffn_ln_blocks": [ctypes.c_long],
    "pcm_ffn_ln_forward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _P, ctypes.c_uint, ctypes.c_uint,
                               _P, _P, _P, _P, _P, _P],
    "pcm_ffn_ln_backward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, ctypes.c_uint,
                                _P, _P, _P, _P, _P, _P],
    "pcm_ffn_ln_forward2_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _P, ctypes.c_uint, ctypes.c_uint,
                                _P, _P, _P, _P, _P, _P, ctypes.c_long, _P, _P, _P],
    "pcm_ffn_ln_backward2_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, ctypes.c_uint,
                                 _P, _P, _P, _P, _P, _P],
    "pcm_ffn_ln_mfma_supported": [_i, _i],
    "pcm_ffn_ln_mfma_blocks": [ctypes.c_long],
    "pcm_ffn_ln_mfma_forward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _P, ctypes.c_uint, ctypes.c_uint,
                                    _P, _P, _P, _P, _P, _P, ctypes.c_long, _P, _P, _P],
    "pcm_ffn_ln_mfma_backward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, ctypes.c_uint,
                                     _P, _P, _P, _P, _P, _P],
    "pcm_ffn_reduce_rows_hip": [_i, _i, _P, _P, _P],
    "pcm_ddpm_step_hip": [ctypes.c_long, _i, _P, _P, _P, _P, _P, _f, _f, _f, _f, _f, _f, _P, _P],
    "pcm_gn_mish_supported": [_i, _i, _i],
    "pcm_gn_mish_forward_hip": [_i, _i, _i, _i, _i, _P, _P, _P, _f, _i, _i, _P, _i, _P, _P, _P, _P, _P, _P],
    "pcm_gn_mish_backward_hip": [_i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_im2col_cl_hip": [_i, _i, _i, _i, _i, _i, _i, _P, _i, _P, _P],
    "pcm_col2im_cl_hip": [_i, _i, _i, _i, _i, _i, _i, _P, _i, _P, _P],
    "pcm_bn_relu_supported": [ctypes.c_long, _i],
    "pcm_bn_relu_slots": [ctypes.c_long, _i],
    "pcm_bn_relu_forward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _f, _f, _P, _P, _i, _P, _P, _P, _P, _P],
    "pcm_bn_relu_backward_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P, _P, _P, _i, ctypes.c_double, _P],
    "pcm_bn_act_forward_hip": [ctypes.c_long, _i, _i, _i, _P, _P, _P, _f, _f, _P, _P, _i, _P, _P, _P, _P, _P],
    "pcm_bn_act_backward_hip": [ctypes.c_long, _i, _i, _i, _P, _P, _P, _P, _P, _P, _i, ctypes.c_double, _P],
    "pcm_voxel_keys_hip": [_i, _i, _P, _P, ctypes.c_double, _P, _P, _P, _P, _P],
    "pcm_add_cast2_hip": [ctypes.c_long, ctypes.c_long, _P, _P, _P, _P, _P],
    "pcm_add2_cast_hip": [ctypes.c_long, _P, _P, _P, _P],
    "pcm_add2_cast2_hip": [ctypes.c_long, _P, _P, _P, _P, _P],
    "pcm_add3_cast2_hip": [ctypes.c_long, _P, _P, _P, _P, _P, _P],
    "pcm_add4_cast2_hip": [ctypes.c_long, _P, _P, _P, _P, _P, _P, _P],
    "pcm_coord_embed_sine_hip": [ctypes.c_long, _i, _i, _P, _P, _P, _P],
    "pcm_act_loss_forward_hip": [_i, _i, _i, _i, _i, _P, _P, _P, _i, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P],
    "pcm_act_loss_backward_hip": [_i, _i, _P, _P, _P, ctypes.c_float, _P, _P, _P, _i, _P, _i, _P, _P, _P],
    "pcm_cvae_latent_forward_hip": [_i, _i, _i, _P, _P, _P, ctypes.c_uint, _P, _P, _P, _P, _P, _P],
    "pcm_cvae_latent_backward_hip": [_i, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_colsum_slots": [ctypes.c_long, _i],
    "pcm_slab_sum_hip": [_i, ctypes.c_long, _P, _i, _P, _P],
    "pcm_reduce_batch_hip": [_i, _P, _P, _P, _P, _P, _P, _P],
    "pcm_colsum_batch_hip": [_i, _P, _P, _P, _P, _P, _P, _P, _P],
    "pcm_copy_batch_hip": [_i, _P, _P, _P, _P],
    "pcm_incr_i64_batch_hip": [_i, _P, _P],
    "pcm_colsum_hip": [ctypes.c_long, _i, _i, _i, _P, ctypes.c_long, _P, ctypes.c_long, _P, ctypes.c_long, _P, _i, _P, _P],
    "pcm_attn_small_supported": [_i, _i, _i],
    "pcm_attn_small_forward_hip": [_i, _i, _i, _i, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, _f, _f, _P, ctypes.c_uint, _P, _P, _P],
    "pcm_attn_small_backward_hip": [_i, _i, _i, _i, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, _f, _f, _P, ctypes.c_uint,
                                    _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P],
    "pcm_attn_flash_supported": [_i, _i, _i],
    "pcm_attn_flash_forward_hip": [_i, _i, _i, _i, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, _f, _f, _P, ctypes.c_uint, _P, _P, _P],
    "pcm_attn_flash_backward_hip": [_i, _i, _i, _i, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, _f, _f, _P, ctypes.c_uint,
                                    _P, _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P],
    "pcm_attn_flash_backward_stages_hip": [_i, _i, _i, _i, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, _f, _f, _P, ctypes.c_uint,
                                           _P, _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _i, _P],
    "pcm_graph_replace_memsets": [_P, _P],
    "pcm_hip_runtime_version": [_P],
    "pcm_memset_async": [_P, ctypes.c_int, ctypes.c_long, ctypes.c_int, _P],
    "pcm_optim_partials_capacity": [],
    "pcm_grad_sumsq_hip": [ctypes.c_long, _P, _P, _P, _P],
    "pcm_adamw_flat_hip": [ctypes.c_long, _P, _P, _P, _P, _P, _P, _i, _P, _P, _P],
    "pcm_xfer_batch_hip": [_i, _P, _P, _P, _P, _P],
}

# functions that return a size (long); everything else returns an int status
LONG_RESULTS = ("pcm_scatter_plan_ws_ints", "pcm_scatter_plan_sorted_scratch_ints", "pcm_sa_index_det_scratch_ints",
                "pcm_sa_bwd1_det_ws_bytes", "pcm_ball_query_ws_bytes")

_LIB = None


class PointopsLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP sources for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU.  Two targets: the shipped library
    (lib/) and `next` (lib_next/: round 5's untimed rewrites of ten files, selected only by PCM_POINTOPS_LIB for the A/B on hardware)."""
    import sys

    r = subprocess.run(["make", "-C", CSRC_DIR, "-j8", "all", "next"], stdout=None if verbose else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    # the device-only subtarget feature of csrc/Makefile (NO_PK) makes the HOST half of every hipcc call print
    # "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)": expected, dropped here
    noise = "is not a recognized feature for this target"
    err = "\n".join(line for line in r.stderr.splitlines() if noise not in line)
    if err.strip():
        print(err, file=sys.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, r.args)
    return LIB_PATH


def load():
    """Load the library and bind every declared symbol.  Raises PointopsLibraryError if anything
    is missing -- the product never degrades to a non-HIP path."""
    global _LIB, LIB_PATH
    if _LIB is not None:
        return _LIB
    # PCM_POINTOPS_LIB: another BUILD of the same library (the address-sanitizer build of `make -C csrc asan`, tools/run_asan.sh).
    # Still the HIP library with every symbol checked below -- not a fallback path.
    LIB_PATH = os.environ.get("PCM_POINTOPS_LIB", LIB_PATH)
    if not os.path.exists(LIB_PATH):
        raise PointopsLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC_DIR}` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    if hasattr(lib, "wavesim_stats"):
        # tests/wavesim's host model exports the same entry points for HOST pointers: test infrastructure, never a product path
        raise PointopsLibraryError(f"{LIB_PATH} is the host wave64 model of tests/wavesim (test infrastructure): the product loads the "
                                   "gfx950 build only. There is no CPU fallback.")
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PointopsLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = args
        fn.restype = ctypes.c_long if name in LONG_RESULTS else ctypes.c_int
    lib.pcm_version.restype = ctypes.c_char_p
    lib.pcm_version.argtypes = []
    _LIB = lib
    return lib


def on_hip(device):
    """Is `device` (a torch.device) the HIP device?  One place for the question the trainer and the flat optimizer ask before they
    take the library path -- the product has no CPU path, they raise otherwise."""
    return device.type == "cuda"


def raw_stream():
    """hipStream_t (as an int) of torch's current stream on the current device.  The public route,
    `torch.cuda.current_stream().cuda_stream`, builds a Stream object and costs ~13 us per call -- 0.2 ms per training step
    over the eager launches of hybrid mode (tools/dbg/host_profile.py); this is one C call."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def copy_batch(pairs):
    """dst.copy_(src) for every (dst, src) pair of same-shape, same-dtype, contiguous tensors on ONE device, as one launch
    per 32 (csrc/tokens.hip pcm_copy_batch_hip) on the current stream; anything else falls back to Tensor.copy_."""
    import torch

    fast, dev = [], None
    for d, s in pairs:
        if (d.is_cuda and s.is_cuda and d.device == s.device and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous()
                and s.is_contiguous() and (dev is None or d.device == dev)):
            dev = d.device
            if d.numel():
                fast.append((d, s))
        else:
            d.copy_(s, non_blocking=True)
    if not fast:
        return
    n = len(fast)
    P, Lg = ctypes.c_void_p * n, ctypes.c_long * n
    with torch.cuda.device(dev):
        rc = load().pcm_copy_batch_hip(n, P(*[d.data_ptr() for d, _ in fast]), P(*[s.data_ptr() for _, s in fast]),
                                       Lg(*[d.numel() * d.element_size() for d, _ in fast]), raw_stream())
    check(rc, "pcm_copy_batch_hip")


XFER_ZERO, XFER_SET_BF16, XFER_SET_F32, XFER_ADD_BF16, XFER_ADD_F32, XFER_COPY_2B = range(6)


def xfer_batch(jobs):
    """jobs: list of (dst_ptr, src_ptr, numel, kind) on the current device -> csrc/optim.hip pcm_xfer_batch_hip on the current stream
    (96 jobs per launch, tables by value: capturable)."""
    n = len(jobs)
    if not n:
        return
    P, Lg, I = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    rc = load().pcm_xfer_batch_hip(n, P(*[j[0] for j in jobs]), P(*[j[1] or None for j in jobs]), Lg(*[j[2] for j in jobs]),
                                   I(*[j[3] for j in jobs]), raw_stream())
    check(rc, "pcm_xfer_batch_hip")


def copy_pairs(pairs):
    """dst.copy_(src) for every (dst, src) pair of one dtype each: contiguous same-dtype pairs on one HIP device ride pcm_xfer_batch_hip
    (one launch per 96, 8192-element chunks: the framework's multi-tensor copy moves the same bytes at 2.2 TB/s), anything else goes
    through torch._foreach_copy_.  Results are plain copies either way."""
    import torch

    fast, rest, dev = [], {}, None
    for d, s in pairs:
        es = d.element_size()
        if (d.is_cuda and s.is_cuda and d.device == s.device and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous()
                and s.is_contiguous() and es in (2, 4) and (dev is None or d.device == dev) and d.numel() * es // 2 < (1 << 31)):
            dev = d.device
            if d.numel():
                fast.append((d.data_ptr(), s.data_ptr(), d.numel() * es // 2, XFER_COPY_2B))
        else:
            rest.setdefault((d.dtype, s.dtype, d.device), []).append((d, s))
    if fast:
        with torch.cuda.device(dev):
            xfer_batch(fast)
    for grp in rest.values():
        torch._foreach_copy_([d for d, _ in grp], [s for _, s in grp])


def check(rc, name):
    if rc != 0:
        if rc >= 1000:
            raise PointopsLibraryError(f"{name}: HIP error {rc - 1000}")
        raise PointopsLibraryError(f"{name}: rejected arguments (status {rc})")
