"""hipGraph capture helpers.

``captured(fn)`` captures ``fn()`` into a hipGraph the way the trainer and the rollout wrapper need it and returns
``(graph, result)``.  Between capture and instantiation every MEMSET node is replaced by a fill-kernel node
(csrc/graph_fix.hip): memset nodes created by stream capture replay with a garbage pattern on this ROCm release, and
PyTorch's reductions depend on one to zero their inter-block semaphores -- a replayed training step otherwise produces
stale or non-finite bias gradients after a few dozen replays (measured; tools/dbg/graph_memset2.py, graph_reduce4.py).
"""
import ctypes

import torch

from . import _lib


def new_graph():
    """A graph object that keeps its hipGraph_t after capture, so it can be patched before it is instantiated."""
    return torch.cuda.CUDAGraph(keep_graph=True)


def finalize(graph):
    """Patch + instantiate a freshly captured graph; returns the number of memset nodes that were replaced."""
    n = ctypes.c_int(0)
    rc = _lib.load().pcm_graph_replace_memsets(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(n))
    _lib.check(rc, "pcm_graph_replace_memsets")
    graph.instantiate()
    return n.value


def captured(fn, pool=None):
    """Capture ``fn()`` (thread-local error mode: RCCL's watchdog / other host threads may touch the HIP API meanwhile)."""
    graph = new_graph()
    kw = {} if pool is None else {"pool": pool}
    with torch.cuda.graph(graph, capture_error_mode="thread_local", **kw):
        result = fn()
    graph.memset_nodes_replaced = finalize(graph)
    return graph, result
