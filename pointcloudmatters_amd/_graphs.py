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


_NEEDS_FIX = {}  # device index -> does this runtime replay captured memset nodes wrongly?  (decided once per process)


def memset_fix_needed(device=None):
    """One-time self-test, per device: capture raw hipMemset(D32)Async nodes of several sizes / patterns WITHOUT the rewrite,
    replay each graph eight times and compare with the pattern.  True (rewrite every captured memset node) if any replay is
    wrong -- the case on ROCm 7.2, where the nodes replay a corrupted pattern from the second launch on --, False on a runtime
    where they replay correctly, so the workaround retires itself.  PCM_GRAPH_MEMSET_FIX=1 / 0 overrides the test."""
    import os

    forced = os.environ.get("PCM_GRAPH_MEMSET_FIX", "auto")
    if forced in ("0", "1"):
        return forced == "1"
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index in _NEEDS_FIX:
        return _NEEDS_FIX[dev.index]
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    bad = False
    with torch.cuda.device(dev):
        for n_ints, d32, value in ((1, False, 0), (16, False, 0), (1024, True, 0), (1000, True, 0x01020304), (77, False, 0x5A)):
            buf = torch.full((n_ints,), 7, device=dev, dtype=torch.int32)
            out = torch.zeros(n_ints, device=dev, dtype=torch.int32)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                st = torch.cuda.current_stream().cuda_stream
                rc = hip.hipMemsetD32Async(buf.data_ptr(), value, n_ints, st) if d32 else hip.hipMemsetAsync(buf.data_ptr(), value, n_ints * 4, st)
                buf.add_(1)
                out.copy_(buf)
            if rc != 0:
                bad = True
                break
            want = (value if d32 else (value & 0xFF) * 0x01010101) + 1
            for _ in range(8):
                g.replay()
                torch.cuda.synchronize(dev)
                if not bool((out == want).all()):
                    bad = True
                    break
            del g
            if bad:
                break
    _NEEDS_FIX[dev.index] = bad
    return bad


def finalize(graph):
    """Patch (if this runtime needs it: memset_fix_needed) + instantiate a freshly captured graph; returns the number of memset
    nodes that were replaced."""
    n = ctypes.c_int(0)
    if memset_fix_needed():
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
