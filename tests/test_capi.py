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
