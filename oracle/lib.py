"""ctypes loader for the C oracle (test infrastructure).  Builds it with gcc on first use."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpcm_oracle.so")
_LIB = None

_F = ctypes.c_void_p  # float*
_I = ctypes.c_void_p  # int*
_i = ctypes.c_int
_f = ctypes.c_float

_SIGS = {
    "pcm_opt_n_threads_cpu": [_i],
    "pcm_oracle_set_threads": [_i],
    "pcm_farthest_point_sampling_cpu": [_i, _i, _F, _I, _I, _F, _I],
    "pcm_knn_query_cpu": [_i, _i, _F, _F, _I, _I, _I, _F],
    "pcm_ball_query_cpu": [_i, _i, _f, _f, _F, _F, _I, _I, _I, _F],
    "pcm_random_ball_query_cpu": [_i, _i, _f, _f, _I, _F, _F, _I, _I, _I, _F],
    "pcm_grouping_forward_cpu": [_i, _i, _i, _F, _I, _F],
    "pcm_grouping_backward_cpu": [_i, _i, _i, _F, _I, _F],
    "pcm_interpolation_forward_cpu": [_i, _i, _i, _F, _I, _F, _F],
    "pcm_interpolation_backward_cpu": [_i, _i, _i, _F, _I, _F, _F],
    "pcm_subtraction_forward_cpu": [_i, _i, _i, _F, _F, _I, _F],
    "pcm_subtraction_backward_cpu": [_i, _i, _i, _I, _F, _F, _F],
    "pcm_aggregation_forward_cpu": [_i, _i, _i, _i, _F, _F, _F, _I, _F],
    "pcm_aggregation_backward_cpu": [_i, _i, _i, _i, _F, _F, _F, _I, _F, _F, _F, _F],
    "pcm_attention_relation_step_forward_cpu": [_i, _i, _i, _F, _F, _F, _I, _I, _F],
    "pcm_attention_relation_step_backward_cpu": [_i, _i, _i, _F, _F, _F, _F, _F, _F, _I, _I, _F],
    "pcm_attention_fusion_step_forward_cpu": [_i, _i, _i, _F, _F, _I, _I, _F],
    "pcm_attention_fusion_step_backward_cpu": [_i, _i, _i, _F, _F, _F, _F, _I, _I, _F],
}


def build(force=False):
    """Compile oracle/libpcm_oracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "pcm_oracle.c")
    if (
        force
        or not os.path.exists(_SO)
        or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(src[:-2] + ".h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpcm_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_sanitized():
    """oracle/libpcm_oracle_asan.so: the same source with -fsanitize=address,undefined (tests/test_oracle_sanitized.py)."""
    subprocess.check_call(["make", "-C", _HERE, "libpcm_oracle_asan.so"], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "libpcm_oracle_asan.so")


def load():
    global _LIB
    if _LIB is None:
        so = os.environ.get("PCM_ORACLE_LIB")  # another build of the same source (the sanitized one)
        if not so:
            so = build()
        lib = ctypes.CDLL(so)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        _LIB = lib
    return _LIB
