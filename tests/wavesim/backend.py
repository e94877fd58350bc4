"""TEST-ONLY: run the product's Python layer (pointcloudmatters_amd/pointops/*.py, policy/sa_fused.py ...) on HOST tensors against
libpcm_wavesim.so -- the product's kernel SOURCES compiled for the CPU wave64 model (tests/wavesim/build.py).

The product itself has no CPU path and keeps none: every op rejects non-HIP tensors and `_lib.load()` raises without the gfx950
library.  `simulated_device()` substitutes, for the duration of a test and by monkeypatching from the OUTSIDE:
  * `_lib.load()`            -> the host-model library, bound with the same ctypes signatures (symbols of files that are not modelled
                                are simply absent: using one raises AttributeError);
  * the HIP-tensor check, the raw-stream getter and `torch.cuda.device(<cpu>)` -> no-ops.
What this shows: the logic of the kernels and of the wrappers above them (bit-exact against the oracle where the GPU tests demand it).
What it cannot show: anything about the hardware (timing, caches, races between lanes, the packed-fp32 hazard): tests -m gpu."""
import contextlib
import ctypes
import sys

import torch

_SIM = None


def library():
    global _SIM
    if _SIM is None:
        from pointcloudmatters_amd import _lib
        from tests.wavesim import build

        lib = ctypes.CDLL(build.build())
        for name, args in _lib.SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.argtypes = args
                fn.restype = ctypes.c_long if name in _lib.LONG_RESULTS else ctypes.c_int
        lib.wavesim_stats.argtypes = [ctypes.c_void_p]
        _SIM = lib
    return _SIM


def stats():
    out = (ctypes.c_long * 3)()
    library().wavesim_stats(out)
    return {"switches": out[0], "cross_lane_ops": out[1], "barriers": out[2]}


@contextlib.contextmanager
def simulated_device(claim_cuda=False):
    """claim_cuda: additionally make every tensor answer `is_cuda == True` (a property put on the Python class torch.Tensor for the
    duration): the policy-level fused ops decide with `x.is_cuda` whether their kernels apply.  Data stays in host memory."""
    from pointcloudmatters_amd import _lib
    from pointcloudmatters_amd.pointops import _common

    sim = library()
    saved = []

    def patch(obj, name, value):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    real_device = torch.cuda.device

    class _NoDevice(contextlib.nullcontext):
        pass

    def device(d):
        dev = torch.device(d) if not isinstance(d, torch.device) else d
        return _NoDevice() if dev.type == "cpu" else real_device(d)

    patch(_lib, "_LIB", sim)
    patch(_lib, "load", lambda: sim)
    patch(_common, "require_hip", lambda *t: None)
    patch(torch.cuda, "device", device)
    patch(torch.cuda, "is_current_stream_capturing", lambda: False)
    patch(torch.cuda, "synchronize", lambda *a, **k: None)

    class _Event:  # one synchronous stream: every event has already happened
        def __init__(self, *a, **k): pass
        def record(self, *a, **k): pass
        def wait(self, *a, **k): pass
        def synchronize(self): pass
        def query(self): return True
        def elapsed_time(self, other): return 0.0

    class _Stream:
        cuda_stream = 0
        device = torch.device("cpu")
        def __init__(self, *a, **k): pass
        def wait_event(self, e): pass
        def wait_stream(self, s): pass
        def synchronize(self): pass
        def record_event(self, e=None): return e if e is not None else _Event()
        def query(self): return True
        def __enter__(self): return self
        def __exit__(self, *exc): return False

    one = _Stream()
    patch(torch.cuda, "current_stream", lambda *a, **k: one)
    patch(torch.cuda, "default_stream", lambda *a, **k: one)
    patch(torch.cuda, "Stream", _Stream)
    patch(torch.cuda, "Event", _Event)
    patch(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    patch(torch.cuda, "current_device", lambda: 0)
    patch(torch.Tensor, "record_stream", lambda self, stream: None)

    # torch.autocast("cuda") is switched off by torch itself on a host without a device.  The product's fused nodes ask
    # torch.is_autocast_enabled("cuda") / get_autocast_dtype("cuda") and cast by hand, everything else relies on the dispatcher's
    # autocast: both are served from the HOST autocast state while the model is active ("cuda" regions become "cpu" regions).
    real_autocast, real_enabled, real_dtype = torch.autocast, torch.is_autocast_enabled, torch.get_autocast_dtype

    class _Autocast(real_autocast):
        def __init__(self, device_type, *a, **k):
            super().__init__("cpu" if device_type == "cuda" else device_type, *a, **k)

    # torch.mm / bmm(..., out_dtype=fp32) (bf16 operands, fp32 result) exist for the device backend only: same values on the host
    real_mm, real_bmm = torch.mm, torch.bmm

    def mm(a, b, out_dtype=None, **k):
        return real_mm(a, b, **k) if out_dtype is None else real_mm(a.to(out_dtype), b.to(out_dtype), **k)

    def bmm(a, b, out_dtype=None, **k):
        return real_bmm(a, b, **k) if out_dtype is None else real_bmm(a.to(out_dtype), b.to(out_dtype), **k)

    patch(torch, "mm", mm)
    patch(torch, "bmm", bmm)
    patch(torch, "autocast", _Autocast)
    patch(torch, "is_autocast_enabled", lambda device_type=None: real_enabled("cpu" if device_type in (None, "cuda") else device_type))
    patch(torch, "get_autocast_dtype", lambda device_type: real_dtype("cpu" if device_type == "cuda" else device_type))
    for mod in list(sys.modules.values()):
        name = getattr(mod, "__name__", "")
        if name.startswith("pointcloudmatters_amd") and hasattr(mod, "_raw_stream"):
            patch(mod, "_raw_stream", lambda: 0)
    patch(_lib, "raw_stream", lambda: 0)
    if claim_cuda:
        patch(_lib, "on_hip", lambda device: True)  # the trainer / flat optimizer take the library path on host tensors
        torch.Tensor.is_cuda = property(lambda self: True)
    try:
        yield torch.device("cpu")
    finally:
        if claim_cuda:
            del torch.Tensor.is_cuda  # the C base class's descriptor is visible again
        for obj, name, value in reversed(saved):
            setattr(obj, name, value)
