"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pcm_pointops.h
declares (no compute calls -- there is no GPU here); the product refuses CPU tensors loudly."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcm_pointops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_sixteen_reference_launchers():
    names = declared_symbols()
    for op in ("farthest_point_sampling", "knn_query", "ball_query", "random_ball_query", "grouping_forward",
               "grouping_backward", "interpolation_forward", "interpolation_backward", "subtraction_forward",
               "subtraction_backward", "aggregation_forward", "aggregation_backward",
               "attention_relation_step_forward", "attention_relation_step_backward",
               "attention_fusion_step_forward", "attention_fusion_step_backward"):
        assert f"pcm_{op}_hip" in names  # pointops_api.cpp:16-31


def test_library_builds_and_exports_every_declared_symbol():
    from pointcloudmatters_amd import _lib

    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    bound = _lib.load()
    assert b"gfx950" in bound.pcm_version()
    assert set(_lib.SIGNATURES) <= set(declared_symbols())
    # host-only entry point: the reference's block-size rule
    for n, want in ((1, 1), (63, 32), (64, 64), (1000, 512), (1024, 1024), (4096, 1024)):
        assert bound.pcm_opt_n_threads(n) == want


def test_product_refuses_cpu_tensors():
    import pointcloudmatters_amd.pointops as po
    from pointcloudmatters_amd._lib import PointopsLibraryError

    xyz = torch.zeros(8, 3)
    off = torch.tensor([8])
    with pytest.raises(PointopsLibraryError):
        po.farthest_point_sampling(xyz, off, torch.tensor([4]))
    with pytest.raises(PointopsLibraryError):
        po.knn_query(4, xyz, off)
    with pytest.raises(PointopsLibraryError):
        po.grouping(torch.zeros(8, 4, dtype=torch.int32), torch.zeros(8, 2), xyz)


def test_public_api_names_match_reference():
    import pointcloudmatters_amd.pointops as po

    # /root/reference/libs/pointops/functions/__init__.py:1-14
    for name in ("aggregation", "attention_fusion_step", "attention_relation_step", "grouping", "grouping2",
                 "interpolation", "interpolation2", "ball_query", "knn_query", "random_ball_query",
                 "farthest_point_sampling", "subtraction", "ball_query_and_group", "batch2offset",
                 "knn_query_and_group", "offset2batch", "query_and_group"):
        assert callable(getattr(po, name))


def test_host_side_argument_checks_need_no_gpu():
    """Entry points validate their arguments before any launch: the rejections (and the empty calls) are testable here."""
    from pointcloudmatters_amd import _lib

    L = _lib.load()
    P, Lg, I = ctypes.c_void_p * 1, ctypes.c_long * 1, ctypes.c_int * 1
    fake = 0x1000  # never dereferenced: the calls below are rejected first
    assert L.pcm_xfer_batch_hip(0, None, None, None, None, None) == 0
    assert L.pcm_xfer_batch_hip(-1, None, None, None, None, None) == 1
    assert L.pcm_xfer_batch_hip(1, P(fake), P(fake), Lg(16), I(6), None) == 1      # unknown kind
    assert L.pcm_xfer_batch_hip(1, P(fake), P(None), Lg(16), I(_lib.XFER_SET_BF16), None) == 1  # missing source
    assert L.pcm_xfer_batch_hip(1, P(None), P(fake), Lg(16), I(_lib.XFER_ZERO), None) == 1      # missing destination
    assert L.pcm_xfer_batch_hip(1, P(fake), P(fake), Lg(-4), I(_lib.XFER_ZERO), None) == 1
    assert L.pcm_xfer_batch_hip(1, P(fake), P(fake), Lg(1 << 31), I(_lib.XFER_ZERO), None) == 1  # job too long for one table entry
    assert L.pcm_ffn_ln_mfma_supported(512, 32) == 1 and L.pcm_ffn_ln_mfma_supported(384, 32) == 0
    assert L.pcm_ffn_ln_mfma_blocks(4120) == 258 and L.pcm_ffn_ln_mfma_blocks(1) == 1  # one partial row per 16-row tile
    assert L.pcm_ball_query_ws_bytes(512) == 0 and L.pcm_ball_query_ws_bytes(65536) > 0


def _prototypes():
    """(name, [argument declarations]) of every `int pcm_*_hip(...)` the header declares (comments stripped)."""
    text = open(os.path.join(ROOT, "include", "pcm_pointops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for name, arglist in re.findall(r"^int (pcm_\w+_hip)\(([^;]*?)\);", text, flags=re.M | re.S):
        out.append((name, [" ".join(a.split()) for a in arglist.split(",")]))
    return out


def _call_with_sizes(fn, decls, size):
    """Every pointer NULL, every float 1.0, every integer argument = `size`."""
    args = []
    for d in decls:
        if "*" in d:
            args.append(None)
        elif d.startswith("float"):
            args.append(ctypes.c_float(1.0))
        elif d.startswith("double"):
            args.append(ctypes.c_double(1.0))
        elif d.startswith("long") or d.startswith("unsigned long"):
            args.append(ctypes.c_long(size))
        else:
            args.append(ctypes.c_int(size))
    fn.restype, fn.argtypes = ctypes.c_int, None
    return fn(*args)


def test_product_loader_refuses_the_host_model_library():
    """PCM_POINTOPS_LIB selects another BUILD of the gfx950 library (the sanitizer build); pointed at tests/wavesim's host model -- the same
    entry points for host pointers -- the loader refuses: the model is test infrastructure, the product has no CPU path."""
    import subprocess
    import sys

    from tests.wavesim import build

    if not os.path.exists(build.CLANG):
        pytest.skip("needs the ROCm clang++")
    so = build.build()
    code = ("import sys; sys.path.insert(0, %r)\nfrom pointcloudmatters_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.PointopsLibraryError as e:\n    print('REFUSED', 'host wave64 model' in str(e))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCM_POINTOPS_LIB=so), capture_output=True, text=True, timeout=300)
    assert "REFUSED True" in out.stdout, out.stdout + out.stderr


def test_runtime_version_and_memset_entry_points_need_no_gpu():
    """pcm_hip_runtime_version / pcm_memset_async (csrc/graph_fix.hip): what _graphs.memset_fix_needed decides on.  The version is the
    runtime the LIBRARY is bound to -- in a process that imported torch first, the copy torch ships -- and anything up to 7.2.x keeps
    the captured-memset rewrite on without a probe."""
    from pointcloudmatters_amd import _graphs, _lib

    L = _lib.load()
    v = _graphs.runtime_version()
    assert 6 * 10000000 <= v < 99 * 10000000
    assert L.pcm_hip_runtime_version(None) == 1
    assert L.pcm_memset_async(None, 0, -1, 0, None) == 1 and L.pcm_memset_async(None, 0, 0, 1, None) == 0
    assert L.pcm_memset_async(None, 0, 4, 0, None) == 1  # missing destination
    assert _graphs.KNOWN_BAD_UP_TO == 70299999


def test_every_entry_point_rejects_negative_sizes_and_takes_empty_calls_without_the_runtime():
    """The status contract of include/pcm_pointops.h over EVERY launcher it declares (the reference's launchers return void and
    run into undefined behaviour on such arguments): negative sizes are PCM_ERR_BAD_ARG / PCM_ERR_UNSUPPORTED, never PCM_OK; an
    all-zero call is either empty (PCM_OK) or rejected (nsample = 0 ...), and neither asks the HIP runtime for anything -- there is
    no device here, so a status >= PCM_ERR_HIP_BASE would show that it did.  No pointer is dereferenced on these paths."""
    from pointcloudmatters_amd import _lib

    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    protos = _prototypes()
    assert len(protos) >= 70 and {"pcm_farthest_point_sampling_hip", "pcm_knn_query_hip", "pcm_xfer_batch_hip"} <= {n for n, _ in protos}
    bad = []
    for name, decls in protos:
        neg = _call_with_sizes(getattr(lib, name), decls, -1)
        zero = _call_with_sizes(getattr(lib, name), decls, 0)
        if neg not in (1, 2) or zero not in (0, 1, 2):
            bad.append((name, neg, zero))
    assert not bad, bad


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI, not a C++ one: include/pcm_pointops.h compiles as strict C99 (and as C++17), and a C program that
    takes the address of EVERY declared function links against libpcm_pointops.so with gcc alone and runs the host-only entry points
    (no torch, no hipcc, no device): what a cgo / JNI / ctypes binding on the reference's side relies on."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not found")
    from pointcloudmatters_amd import _lib

    _lib.build()
    inc, libdir = os.path.join(ROOT, "include"), os.path.dirname(_lib.LIB_PATH)
    names = declared_symbols()
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "pcm_pointops.h"\n'
        "typedef void (*fn_t)(void);\n"
        "static const fn_t table[] = {\n" + "".join(f"    (fn_t){n},\n" for n in names) + "};\n"
        "int main(void) {\n"
        "    size_t i, n = sizeof table / sizeof table[0];\n"
        "    for (i = 0; i < n; ++i) if (!table[i]) return 2;\n"
        '    if (!strstr(pcm_version(), "gfx950")) return 3;\n'
        "    if (pcm_opt_n_threads(1000) != 512 || pcm_opt_n_threads(4096) != 1024) return 4;\n"
        "    if (pcm_knn_query_hip(-1, 16, 0, 0, 0, 0, 0, 0, 0) != PCM_ERR_BAD_ARG) return 5;\n"
        "    if (pcm_farthest_point_sampling_hip(0, 0, 0, 0, 0, 0, 0, 0) != PCM_OK) return 6;\n"
        '    printf("%u symbols\\n", (unsigned)n);\n    return 0;\n}\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, str(src), "-o", str(exe),
                           "-L", libdir, "-lpcm_pointops", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.strip() == f"{len(names)} symbols"
    cxx = tmp_path / "abi.cpp"
    cxx.write_text('#include "pcm_pointops.h"\nint main() { return pcm_opt_n_threads(64) == 64 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(cxx)])
