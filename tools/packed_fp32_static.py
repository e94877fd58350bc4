#!/usr/bin/env python
"""Static effect of re-enabling packed fp32 per file in lib_next (csrc/Makefile NEXT_PK_FILES): VALU instructions of every kernel of those
files compiled with NO_PK and without, and the packed forms the compiler emits (`op_sel:` = the hazard form of round 4's hardware
reproducer: must be 0).  From the gfx950 assembly; no GPU.      python tools/packed_fp32_static.py > profiles/rNN_packed_fp32_static.md"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", CSRC, "--cuda-device-only", "-S"]
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def kernels(path, packed):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + BASE + ([] if packed else NO_PK) + [path, "-o", out], check=True, capture_output=True)
        lines = open(out).read().splitlines()
    st = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    res = {}
    for k, i in enumerate(st):
        body = lines[i:st[k + 1] if k + 1 < len(st) else len(lines)]
        end = next((j for j, l in enumerate(body) if "s_endpgm" in l), len(body))
        res[lines[i][:-1]] = [x.strip() for x in body[:end] if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))]
    return res


if __name__ == "__main__":
    mk = open(os.path.join(CSRC, "Makefile")).read()
    files = re.search(r"^NEXT_PK_FILES\s*:=\s*(.+)$", mk, re.M).group(1).split()
    print("# Packed fp32 per file in lib_next: static VALU instructions per kernel, NO_PK vs packed (`python tools/packed_fp32_static.py`)\n")
    print("| file | kernel | VALU (NO_PK) | VALU (packed) | change | packed instr. | of them op_sel_hi | op_sel: (hazard form) |")
    print("|---|---|---|---|---|---|---|---|")
    for f in files:
        p = os.path.join(CSRC, "next", f + ".hip")
        a0, a1 = kernels(p, False), kernels(p, True)
        dem = dict(zip(a0, subprocess.run(["c++filt"], input="\n".join(a0), capture_output=True, text=True).stdout.splitlines()))
        for k in a0:
            valu = lambda ins: sum(1 for t in ins if t.startswith("v_") and not t.startswith("v_mfma"))  # noqa: E731
            v0, v1 = valu(a0[k]), valu(a1[k])
            pk = [t for t in a1[k] if re.match(r"v_pk_(add|mul|fma)_f32", t)]
            if not pk:
                continue
            nm = dem[k].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = nm[:nm.index("(")] if "(" in nm else nm
            print("| %s | `%s` | %d | %d | %+.0f %% | %d | %d | %d |" % (f, nm[:70], v0, v1, 100.0 * (v1 - v0) / v0, len(pk), sum("op_sel_hi" in t for t in pk),
                                                                 sum("op_sel:" in t for t in pk)))
