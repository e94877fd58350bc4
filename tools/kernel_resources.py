#!/usr/bin/env python
"""Register / scratch / LDS use of every gfx950 kernel of the library, from the compiler's own metadata (hipcc --cuda-device-only -S, the
Makefile's flags): one row per kernel for the shipped sources (csrc/*.hip) and for the rewrites (csrc/next/*.hip).

    python tools/kernel_resources.py [--md] > profiles/rNN_kernel_resources.md

Used by tests/test_build_flags.py (no kernel spills except the one documented) and to compare a rewrite's register budget with the frozen
kernel's before it is timed (round-5 ADVICE: the SGPR / VGPR / scratch counts of every modified kernel, recorded)."""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang",
         "-packed-fp32-ops", "--cuda-device-only", "-S", "-I", CSRC]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60).stdout.splitlines()
        def short(o):
            o = o.replace("(anonymous namespace)::", "").replace("void ", "")
            depth = 0
            for i in range(len(o) - 1, -1, -1):  # cut the trailing argument list (balanced parentheses)
                depth += o[i] == ")"
                depth -= o[i] == "("
                if depth == 0 and o[i] == "(":
                    return o[:i]
            return o

        return [short(o) for o in out]
    except Exception:  # noqa: BLE001
        return names


def kernels_of(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        flags = FLAGS
        mk = open(os.path.join(CSRC, "Makefile")).read()
        pk_files = re.search(r"^NEXT_PK_FILES\s*:=\s*(.+)$", mk, re.M).group(1).split()
        if os.path.dirname(os.path.abspath(path)) == os.path.join(CSRC, "next") and os.path.basename(path)[:-4] in pk_files:  # built with packed fp32
            flags = [f for f in FLAGS if f not in ("-Xclang", "-target-feature", "-packed-fp32-ops")]
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + [path, "-o", asm], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(path + ": " + r.stderr[-1500:])
        text = open(asm).read()
    rows = []
    for blk in text.split("- .agpr_count:")[1:]:
        def f(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "0"
        rows.append(dict(name=f("name"), vgpr=int(f("vgpr_count")), agpr=int(blk.split("\n")[0].strip() or 0), sgpr=int(f("sgpr_count")),
                         scratch=int(f("private_segment_fixed_size")), lds=int(f("group_segment_fixed_size")),
                         sgpr_spill=int(f("sgpr_spill_count")), vgpr_spill=int(f("vgpr_spill_count"))))
    for row, d in zip(rows, demangle([r_["name"] for r_ in rows])):
        row["short"] = d
    return rows


def collect(paths):
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        return dict(zip(paths, ex.map(kernels_of, paths)))


if __name__ == "__main__":
    shipped = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    nxt = sorted(glob.glob(os.path.join(CSRC, "next", "*.hip")))
    res = collect(shipped + nxt)
    print("# gfx950 kernel resources (compiler metadata; `python tools/kernel_resources.py`)\n")
    print("VGPRs (+ AGPRs) per lane, SGPRs, scratch bytes per lane, static LDS bytes, spilled SGPRs / VGPRs.  Files with a rewrite in `next/`: shipped | next.\n")
    for p in shipped:
        base = os.path.basename(p)
        alt = os.path.join(CSRC, "next", base)
        rows = {r["short"]: r for r in res[p]}
        arows = {r["short"]: r for r in res.get(alt, [])}
        print("## %s%s\n" % (base, " (shipped | next)" if arows else ""))
        print("| kernel | VGPR | SGPR | scratch | LDS | spills s/v |" + (" next: VGPR | SGPR | scratch | spills s/v |" if arows else ""))
        print("|---|---|---|---|---|---|" + ("---|---|---|---|" if arows else ""))
        for k in sorted(set(rows) | set(arows)):
            a, b = rows.get(k), arows.get(k)
            cell = lambda r: "%d%s | %d | %d | %d | %d/%d" % (r["vgpr"], "+%d" % r["agpr"] if r["agpr"] else "", r["sgpr"], r["scratch"], r["lds"], r["sgpr_spill"], r["vgpr_spill"]) if r else "- | - | - | - | -"  # noqa: E731
            cell2 = lambda r: "%d%s | %d | %d | %d/%d" % (r["vgpr"], "+%d" % r["agpr"] if r["agpr"] else "", r["sgpr"], r["scratch"], r["sgpr_spill"], r["vgpr_spill"]) if r else "- | - | - | -"  # noqa: E731
            print("| `%s` | %s |%s" % (k[:100], cell(a), (" %s |" % cell2(b)) if arows else ""))
        print()
