"""Build container only: the committed fixtures ARE what the reference's own Python produces -- tests/golden/make_golden.py re-run into a
scratch directory gives every array of every .npz back (bit for bit on the machine that wrote them; tools/check_golden_regen.py).  Skipped where /root/reference
does not exist (the GPU box); nothing here touches the product path."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (build container only)")
def test_fixtures_regenerate_bit_for_bit_from_the_reference():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_golden_regen.py")], capture_output=True, text=True,
                         timeout=900)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:]
    assert out.returncode == 0, tail
    n, eq = int(tail.split()[0]), int(tail.split()[2])
    # round 4, on the machine that wrote them: 2159 arrays over 16 fixtures, ALL bit-equal.  Elsewhere the float arrays may differ in the last
    # bits (other GEMM kernels); the checker's exit code holds them to 2e-5 of each array's scale and every integer / index array to equality
    assert n >= 2100 and eq >= n // 2, tail
