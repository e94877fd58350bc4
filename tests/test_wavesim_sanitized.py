"""The product's KERNEL SOURCES under AddressSanitizer + UndefinedBehaviorSanitizer, on the host wave64 model.

Device-side sanitizers cannot run on this pool (the boxes are not xnack-enabled: profiles/r03_asan_attempt.log).  tests/wavesim/ compiles
the same .hip files for the CPU; with WAVESIM_SANITIZE=1 they are instrumented, so a load or store one element outside a global buffer, a
static LDS array or the dynamic LDS block, a signed overflow in an index computation or an out-of-range shift inside a kernel aborts the
child process.  Here: the index kernels, the fused SA layer, segmented sums and the round-5 matrix-core kernels over their parity tests (the
whole adopted module -- 451 tests -- was run this way once: profiles/r05_host_model_gpu_suite.md).  Memory safety of the logic, not of the
hardware execution."""
import os
import subprocess
import sys

import pytest

from tests.wavesim import build as _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(_build.CLANG) or _build.asan_runtime() is None, reason="needs the ROCm clang++ and its asan runtime")
def test_kernel_sources_pass_their_parity_tests_under_asan_ubsan():
    env = dict(os.environ, WAVESIM_SANITIZE="1", LD_PRELOAD=_build.asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("PCM_WAVESIM_FULL", None)
    sel = ("fps_bit_exact or knn_bit_exact or ball_query_bit_exact or random_ball or grouping or interpolation or segsum_gpu__ or "
           "sa_fused_gpu__fused_matches or proj_ln_gpu__projection_residual or proj_ln_gpu__linear_from or proj_ln_gpu__consumer or "
           "bn_relu_gpu__bn_without or proj_ln_gpu__backward_chain or proj_ln_gpu__linear_backward")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_wavesim_parity.py", "-k", sel],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    text = r.stdout + r.stderr
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]
    assert r.returncode == 0, text[-3000:]
    tail = r.stdout.strip().splitlines()[-1]
    assert " passed" in tail and int(tail.split()[0]) >= 60, tail
