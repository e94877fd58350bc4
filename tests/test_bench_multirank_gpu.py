"""The complete `bench.py --gpus 2` flow (what the driver launches with torch.distributed.run on a multi-GPU node) on a
one-GPU box: both ranks share device 0 and gloo carries the collectives (PCM_BENCH_SHARE_GPU=1, a test hook in bench.py).
Catches what the numerics tests cannot: ranks falling out of lockstep (a collective issued by rank 0 only hangs the
job), the barrier / max-over-ranks timing, the one-JSON-line contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("dp_mode", ["graph", "hybrid"])
def test_bench_two_ranks_prints_one_line_and_exits(hip_device, dp_mode):
    """dp_mode graph: what the driver's N > 1 runs use since round 4 (the whole step as a chain of hipGraphs cut at the collectives);
    hybrid: the round-3 mode (eager tokenizer), still what ragged batches and the Diffusion Policy take (PCM_DP_MODE selects)."""
    env = dict(os.environ, PCM_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="4", PCM_DP_MODE=dp_mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, res.stdout[-2000:]
    assert len(res.stdout.splitlines()[-1]) < 4096  # the driver keeps a tail of stdout: the line must stay short
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 3 and out["scaling"] == "weak" and out["higher_is_better"]
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2" and out["config"]["step_mode"] == dp_mode
    assert out["config"]["batchnorm"] == "sync"
    ex = out["config"]["gradient_exchange_exposed_ms"]  # per-step exposed exchange time + the four slabs of the flat gradient
    assert ex["steps"] >= 4 and ex["exposed_ms_mean"] >= 0 and len(ex["slabs_mb"]) == 4 and abs(sum(ex["slabs_mb"]) - 96.4) < 1.0
    assert out["value"] > 0 and abs(out["value"] - 16 * 4 / (out["ms_per_step"] * 4 / 1e3)) < 1e-2 * out["value"]
    assert "roofline" in out and "step" in out and "cpu_baseline" not in out  # cpu_baseline: rank 0 at N = 1 only
    assert out["final_loss"] == out["final_loss"]  # finite
