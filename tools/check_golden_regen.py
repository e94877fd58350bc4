"""Build container only: are the committed fixtures what the reference's own Python produces TODAY?

Re-runs tests/golden/make_golden.py (which imports the reference classes from /root/reference) into a scratch directory and compares
every array of every .npz with the committed one: bit-equal, or -- for floating-point arrays -- the largest relative difference
(the generator pins torch to one thread; what is left is library-level reduction order).  Exit code 1 on a missing key / array, a shape
or dtype change, an integer difference, or a float difference above 2e-5 of the array's scale (bit-equal on the machine that wrote the fixtures; another CPU
may pick other GEMM kernels, and the fixture tests themselves hold the product to 1e-4).
    python tools/check_golden_regen.py [fixture names as make_golden.py takes them]
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def compare(old_dir, new_dir, names=None):
    bad, worst, n_arrays, n_equal = [], 0.0, 0, 0
    for fn in sorted(f for f in os.listdir(old_dir) if f.endswith(".npz")):
        if names is not None and fn not in names:
            continue
        a, b = np.load(os.path.join(old_dir, fn), allow_pickle=False), np.load(os.path.join(new_dir, fn), allow_pickle=False)
        if set(a.files) != set(b.files):
            bad.append((fn, "keys differ", sorted(set(a.files) ^ set(b.files))[:5]))
            continue
        for k in a.files:
            x, y = a[k], b[k]
            n_arrays += 1
            if x.shape != y.shape or x.dtype != y.dtype:
                bad.append((fn, k, "shape / dtype", x.shape, y.shape, str(x.dtype), str(y.dtype)))
            elif np.array_equal(x, y, equal_nan=x.dtype.kind == "f"):
                n_equal += 1
            elif x.dtype.kind != "f":
                bad.append((fn, k, "integer / byte arrays differ"))
            else:
                rel = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max() / max(float(np.abs(x).max()), 1e-30))
                worst = max(worst, rel)
                if rel > 2e-5:
                    bad.append((fn, k, "relative difference", rel))
    return bad, worst, n_arrays, n_equal


def main():
    assert os.path.isdir("/root/reference"), "build container only (needs /root/reference)"
    tmp = tempfile.mkdtemp(prefix="pcm_golden_")
    try:
        committed = sorted(f for f in os.listdir(GOLD) if f.endswith(".npz"))
        for f in committed:  # later generators read earlier fixtures (inputs shared between them) ...
            shutil.copy(os.path.join(GOLD, f), tmp)
            os.utime(os.path.join(tmp, f), (0, 0))  # ... but a copy the run did NOT rewrite must not be compared with itself
        env = dict(os.environ, PCM_GOLDEN_OUT=tmp)
        subprocess.check_call([sys.executable, os.path.join(GOLD, "make_golden.py")] + sys.argv[1:], env=env)
        rewritten = {f for f in committed if os.stat(os.path.join(tmp, f)).st_mtime > 1}
        stale = [f for f in committed if f not in rewritten]
        bad, worst, n, eq = compare(GOLD, tmp, rewritten)
        if not sys.argv[1:]:  # a full run must reproduce EVERY committed fixture (a generator that raised early, a conditional write)
            bad += [(f, "not rewritten by the regeneration run") for f in stale]
        elif not rewritten:
            bad.append(("no fixture was rewritten", sys.argv[1:]))
        print(f"{len(rewritten)} of {len(committed)} fixtures regenerated" + (f" (not selected: {', '.join(stale)})" if stale else ""))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{n} arrays: {eq} bit-equal, largest relative difference among the others {worst:.3g}; {len(bad)} beyond 2e-5")
    for b in bad[:20]:
        print("  ", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
