"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no bounds / race tooling).

Device-side ASAN cannot run on this pool (the boxes are not xnack-enabled: profiles/r03_asan_attempt.log), so the part of the
parity chain that CAN be sanitized is: the CPU restatement every index output of the HIP kernels is compared with.  This test
builds oracle/libpcm_oracle_asan.so (`-fsanitize=address,undefined -fno-sanitize-recover`) and runs the oracle's own tests and the
golden-fixture tests against it in a child process with the sanitizer runtime preloaded: a read or write one element outside an
array, a signed overflow in an index computation or a misaligned access in the restatement aborts the child.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime():
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_runtime() is None, reason="gcc's libasan.so not found")
def test_oracle_tests_pass_under_asan_ubsan():
    sys.path.insert(0, ROOT)
    from oracle import lib as oracle_lib

    so = oracle_lib.build_sanitized()
    env = dict(os.environ, LD_PRELOAD=_runtime(), PCM_ORACLE_LIB=so,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle.py", "tests/test_golden_cpu.py",
                        "tests/test_mask_sampling.py", "tests/test_presample.py", "-m", "not gpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error" not in r.stdout + r.stderr, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
