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
KNOWN_BAD_UP_TO = 7 * 10000000 + 2 * 100000 + 99999  # HIP_VERSION encoding: every 7.2.x and older (found on 7.2.26015)


def runtime_version():
    """HIP_VERSION-encoded version of the runtime libpcm_pointops.so is bound to (the one whose graphs are being patched)."""
    v = ctypes.c_int(0)
    _lib.check(_lib.load().pcm_hip_runtime_version(ctypes.byref(v)), "pcm_hip_runtime_version")
    return v.value


def memset_fix_needed(device=None):
    """Rewrite every captured memset node?  PCM_GRAPH_MEMSET_FIX=1 / 0 overrides.  Otherwise:
      * runtimes up to 7.2.x (where the defect was found: replays >= 1 of a captured memset write a stale host pattern): ALWAYS.
        The rewrite costs one kernel node per memset; a probe of five shapes cannot prove a size- or pattern-dependent defect absent,
        and a false negative would corrupt every training graph silently;
      * newer runtimes: a one-time self-test per device -- raw hipMemset(D32)Async nodes of several sizes / patterns captured WITHOUT
        the rewrite, each graph replayed eight times and compared with the pattern -- so that the workaround retires itself once the
        runtime is fixed.  The memsets are issued through libpcm_pointops.so's own binding of the runtime (pcm_memset_async)."""
    import os

    forced = os.environ.get("PCM_GRAPH_MEMSET_FIX", "auto")
    if forced in ("0", "1"):
        return forced == "1"
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index in _NEEDS_FIX:
        return _NEEDS_FIX[dev.index]
    if runtime_version() <= KNOWN_BAD_UP_TO:
        _NEEDS_FIX[dev.index] = True
        return True
    _NEEDS_FIX[dev.index] = bad = memset_self_test(dev)
    return bad


def memset_self_test(dev):
    """True if a captured memset node replays wrongly on `dev` (see memset_fix_needed)."""
    lib = _lib.load()
    bad = False
    with torch.cuda.device(dev):
        for n_ints, d32, value in ((1, False, 0), (16, False, 0), (1024, True, 0), (1000, True, 0x01020304), (77, False, 0x5A)):
            buf = torch.full((n_ints,), 7, device=dev, dtype=torch.int32)
            out = torch.zeros(n_ints, device=dev, dtype=torch.int32)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = lib.pcm_memset_async(ctypes.c_void_p(buf.data_ptr()), value, n_ints if d32 else n_ints * 4, 1 if d32 else 0, st)
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


# ---- a captured step with collectives in it -------------------------------------------------------------------------------
# A data-parallel step contains collectives that must NOT be captured (RCCL owns its streams; the driver's 8-GPU run is the first
# time they meet more than one device): the synchronised BatchNorm statistics of the tokenizer (6 all_gathers forward, 6
# all_reduces backward, configs/trainer/ddp.yaml:9) and the gradient slabs between the backward stages.  Round 3 therefore ran the
# whole tokenizer eagerly at N > 1 ("hybrid": ~95 launches + 12 collectives issued from Python per step, host-paced).  Here the
# step is captured as a CHAIN: capture runs until code reaches a collective, the current graph is closed, the collective is
# recorded as a plain Python call on the tensors the graphs read / write (static addresses in the shared graph pool), and a new
# capture begins behind it.  A replayed step is then  g0 | all_gather | g1 | all_gather | ... | gK  : ~16 graph launches and ~15
# collective launches from the host instead of ~560 kernel launches, every collective still a plain eager RCCL call.
_CHAIN = None


def chain():
    """The SegmentedCapture that is recording right now (None: run collectives inline)."""
    return _CHAIN


def cuts():
    """True while a chain is recording that CUTS the capture at collectives (no stream may stay forked across such a cut)."""
    import os

    return _CHAIN is not None and os.environ.get("PCM_DP_CAPTURE_COLLECTIVES") != "1"


def between(fn):
    """Run `fn()` -- a collective on tensors that live across graph segments -- now, and, when a chain is recording, cut the
    capture around it so that it is re-issued eagerly between the two graphs at every replay."""
    c = _CHAIN
    if c is None:
        return fn()
    return c.between(fn)


class SegmentedCapture:
    def __init__(self, pool=None):
        self.items = []      # ("graph", CUDAGraph) | ("call", fn) in replay order
        self.pool = pool
        self.cur = None
        self.memset_nodes_replaced = 0

    # -- capture ---------------------------------------------------------------------------------------------------------
    def __enter__(self):
        global _CHAIN
        import threading

        assert _CHAIN is None, "segmented captures do not nest"
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        memset_fix_needed()  # the self-test captures graphs of its own: decide before this capture starts
        self._thread = threading.get_ident()
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        self._sctx = torch.cuda.stream(self._stream)
        self._sctx.__enter__()
        # backward nodes must run on THIS thread: a capture begun in thread-local mode has to be ended by the thread that began
        # it, and `between` ends / begins captures from inside backward nodes (the SyncBN gradient statistics)
        self._mt = torch.autograd.set_multithreading_enabled(False)
        self._mt.__enter__()
        _CHAIN = self
        self._begin()
        return self

    def _begin(self):
        g = new_graph()
        kw = {} if self.pool is None else {"pool": self.pool}
        g.capture_begin(capture_error_mode="thread_local", **kw)
        self.cur = g

    def _end(self):
        g, self.cur = self.cur, None
        g.capture_end()
        if self.pool is None:
            self.pool = g.pool()
        self.memset_nodes_replaced += finalize(g)
        self.items.append(("graph", g))

    def between(self, fn):
        import os
        import threading

        if os.environ.get("PCM_DP_CAPTURE_COLLECTIVES") == "1":
            # EXPERIMENT (off by default): leave the collective INSIDE the capture -- RCCL's launch and its stream hand-over become
            # graph nodes, no cut.  Works with a one-rank group on one GPU (tools/dbg/dp_single_rank.py); never run on > 1 device.
            return fn()
        assert threading.get_ident() == self._thread, "a collective was reached on another thread than the capturing one"
        self._end()
        out = fn()  # eager, on the capture stream (not capturing now): pairs up with the other ranks' captures
        self.items.append(("call", fn))
        self._begin()
        return out

    def __exit__(self, et, ev, tb):
        global _CHAIN
        _CHAIN = None
        try:
            if self.cur is not None:
                if et is None:
                    self._end()
                else:  # leave capture mode whatever happened, keep the original exception
                    try:
                        self.cur.capture_end()
                    except Exception:
                        pass
                    self.cur = None
        finally:
            self._mt.__exit__(et, ev, tb)
            self._sctx.__exit__(et, ev, tb)
            torch.cuda.current_stream().wait_stream(self._stream)
        return False

    # -- replay ----------------------------------------------------------------------------------------------------------
    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()

    @property
    def n_graphs(self):
        return sum(1 for k, _ in self.items if k == "graph")

    @property
    def n_calls(self):
        return sum(1 for k, _ in self.items if k == "call")
