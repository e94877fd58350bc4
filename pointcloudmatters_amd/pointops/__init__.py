"""``pointops`` for MI355X -- drop-in for the reference package of the same name.

Exports exactly the names of /root/reference/libs/pointops/functions/__init__.py:1-14 with the same
signatures, dtypes and return values, backed by hand-written HIP kernels for gfx950 behind the C ABI
of ``include/pcm_pointops.h`` (no CUDAExtension, no hipify, no CPU fallback).
"""
from .aggregation import aggregation
from .attention import attention_fusion_step, attention_relation_step
from .grouping import grouping, grouping2
from .interpolation import interpolation, interpolation2
from .query import ball_query, knn_query, random_ball_query
from .sampling import farthest_point_sampling
from .subtraction import subtraction
from .utils import (
    ball_query_and_group,
    batch2offset,
    knn_query_and_group,
    offset2batch,
    query_and_group,
)

__all__ = [
    "aggregation", "attention_fusion_step", "attention_relation_step", "grouping", "grouping2",
    "interpolation", "interpolation2", "ball_query", "knn_query", "random_ball_query",
    "farthest_point_sampling", "subtraction", "ball_query_and_group", "batch2offset",
    "knn_query_and_group", "offset2batch", "query_and_group",
]
