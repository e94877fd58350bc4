"""GPU: hipGraph capture helper (pointcloudmatters_amd/_graphs.py).

Memset nodes created by stream capture replay with a garbage pattern on this ROCm release (first launch correct, later ones
not); PyTorch's multi-block reductions zero their semaphores with one.  ``captured`` replaces them by kernel nodes -- these
tests replay patched graphs a few hundred times with eager work in between and compare every replay with eager results."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    return hip


@pytest.mark.parametrize("n_ints,d32,value", [(1, False, 0), (16, False, 0), (1024, True, 0), (1000, True, 0x01020304), (77, False, 0x5A)])
def test_captured_memsets_replay_correctly(hip_device, n_ints, d32, value):
    from pointcloudmatters_amd._graphs import captured

    hip = _hip()
    buf = torch.full((n_ints,), 7, device=hip_device, dtype=torch.int32)
    out = torch.zeros(n_ints, device=hip_device, dtype=torch.int32)

    def body():
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetD32Async(buf.data_ptr(), value, n_ints, st) if d32 else hip.hipMemsetAsync(buf.data_ptr(), value, n_ints * 4, st)
        assert rc == 0
        buf.add_(1)
        out.copy_(buf)

    graph, _ = captured(body)
    from pointcloudmatters_amd._graphs import memset_fix_needed

    # the rewrite is gated on a one-time self-test of the runtime: needed on ROCm 7.2 (then the node is replaced), a no-op on a
    # runtime whose captured memsets replay correctly -- either way every replay below must be right
    assert graph.memset_nodes_replaced == (1 if memset_fix_needed() else 0)
    want = (value if d32 else (value & 0xFF) * 0x01010101) + 1
    for _ in range(20):
        graph.replay()
        torch.cuda.synchronize()
        assert bool((out == want).all()), out[:8].tolist()


def test_torch_reductions_inside_a_patched_graph(hip_device):
    """Column sums that take ATen's global-reduce path (staging buffer + semaphores zeroed by a captured memset)."""
    from pointcloudmatters_amd._graphs import captured

    torch.manual_seed(0)
    rows, cols = 1024, 512
    x = torch.zeros(rows, cols, device=hip_device, dtype=torch.bfloat16)

    def body():
        a = x * 2.0
        s1 = a.sum(dim=0)
        b = a + 1.0
        s2 = b.sum(dim=0)
        s3 = b.float().sum(dim=0)
        return s1, s2, s3, b.float().sum()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph, res = captured(body)
    from pointcloudmatters_amd._graphs import memset_fix_needed

    assert graph.memset_nodes_replaced >= 1 or not memset_fix_needed(), "expected ATen's semaphore memsets in the captured graph"
    for it in range(1, 400):
        v = float(it % 64) / 64.0
        x.fill_(v)
        y = torch.randn(1024, 1024, device=hip_device) @ torch.randn(1024, 1024, device=hip_device)  # eager work between replays
        _ = y.to(torch.bfloat16).sum(dim=0)
        graph.replay()
        torch.cuda.synchronize()
        s1, s2, s3, tot = res
        assert abs(s1.float().mean().item() - 2 * v * rows) <= 0.01 * rows + 1e-3, it
        assert abs(s3.mean().item() - (2 * v + 1) * rows) <= 1e-2 * rows, it
        assert abs(tot.item() - (2 * v + 1) * rows * cols) <= 1e-2 * rows * cols, it
        assert bool((s2 == s2[0]).all()), it


def test_memset_fix_self_test_is_decided_once_and_overridable(hip_device, monkeypatch):
    from pointcloudmatters_amd import _graphs

    first = _graphs.memset_fix_needed()
    assert isinstance(first, bool) and _graphs._NEEDS_FIX[torch.cuda.current_device()] == first
    assert _graphs.memset_fix_needed() == first  # cached
    version = _graphs.runtime_version()
    assert version >= 6 * 10000000
    if version <= _graphs.KNOWN_BAD_UP_TO:  # the runtimes the defect was found on: no probe can switch the rewrite off
        assert first is True
    observed = _graphs.memset_self_test(hip_device)  # memsets through the library's own runtime binding (pcm_memset_async)
    assert isinstance(observed, bool)
    print(f"captured-memset self-test on HIP runtime {version}: {'replays wrongly' if observed else 'replays correctly'}")
    monkeypatch.setenv("PCM_GRAPH_MEMSET_FIX", "0")
    assert _graphs.memset_fix_needed() is False
    monkeypatch.setenv("PCM_GRAPH_MEMSET_FIX", "1")
    assert _graphs.memset_fix_needed() is True
