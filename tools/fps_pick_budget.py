#!/usr/bin/env python
"""Static issue budget of ONE farthest-point pick: instructions inside the pick loop (the innermost loop that contains the DPP wave
maximum) of the register-resident FPS kernels, from the gfx950 assembly -- shipped csrc/fps.hip (scalar fp32: Makefile NO_PK), csrc/next/fps.hip
compiled the same way (round 5's prologue rewrite: the loop is the same), and csrc/next/fps.hip as it is built (round 6: packed fp32, plain
forms only).  No GPU needed; it says how many instructions a wave must issue per pick, not how long a pick takes.

    python tools/fps_pick_budget.py > profiles/rNN_fps_pick_budget.md"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", CSRC, "--cuda-device-only", "-S"]
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def asm_of(path, packed):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + BASE + ([] if packed else NO_PK) + [path, "-o", out], check=True, capture_output=True)
        return open(out).read()


def pick_loop(asm, kernel_sub):
    lines = asm.splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, i in enumerate(starts):
        if kernel_sub not in lines[i]:
            continue
        body = lines[i:starts[k + 1] if k + 1 < len(starts) else len(lines)]
        best = None
        for h, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if not m:
                continue
            back = [j for j in range(h + 1, len(body)) if re.search(r"s_cbranch\w*\s+%s\b" % re.escape(m.group(1)), body[j])]
            if back:
                ins = [x.split()[0] for x in body[h:back[-1] + 1] if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))]
                if "v_max_u32_dpp" in ins and (best is None or len(ins) < len(best)):
                    best = ins
        return best
    return None


if __name__ == "__main__":
    variants = ((os.path.join(CSRC, "fps.hip"), False, "shipped `csrc/fps.hip` (NO_PK)"),
                (os.path.join(CSRC, "next", "fps.hip"), False, "`next/fps.hip` built with NO_PK (round 5's file)"),
                (os.path.join(CSRC, "next", "fps.hip"), True, "`next/fps.hip` as built: packed fp32, plain forms (round 6)"))
    asms = [(tag, asm_of(p, pk)) for p, pk, tag in variants]
    print("# Instructions a wave issues per farthest-point pick (static, from the gfx950 assembly; `python tools/fps_pick_budget.py`)\n")
    print("| kernel (threads, points per thread) | workload | variant | total | VALU | of which packed | SALU | LDS | barriers |")
    print("|---|---|---|---|---|---|---|---|---|")
    for ks, name, wl in (("pcm_fps_reg_kernelILi128ELi8ELi3ELb1E", "128 x 8", "C2 / C3: 1024 points"), ("pcm_fps_reg_kernelILi256ELi8ELi2ELb1E", "256 x 8", "C4: 2048 points"),
                         ("pcm_fps_reg_kernelILi256ELi16ELi2ELb1E", "256 x 16", "C5: 4096 points"), ("pcm_fps_reg_kernelILi256ELi24ELi2ELb1E", "256 x 24", "REF: ragged ~4096 (<= 6144)")):
        for tag, asm in asms:
            b = pick_loop(asm, ks)
            c = collections.Counter(b)
            fam = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))  # noqa: E731
            print("| `pcm_fps_reg_kernel<%s>` | %s | %s | %d | %d | %d | %d | %d | %d |" % (name, wl, tag, len(b), fam("v_"), fam("v_pk_"), fam("s_"), fam("ds_"), c["s_barrier"]))
