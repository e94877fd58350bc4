"""Static ISA audit: inner loops of the gfx950 device code that keep only one or two global loads in flight before a full
`s_waitcnt vmcnt(0)` -- each iteration is then an exposed L2 / HBM round trip (how round 5 found the A-panel loops of csrc/proj_ln.hip
and the late position-row loads of csrc/drln.hip without a GPU).  Heuristic, for reading -- not a pass/fail gate.

    python tools/isa_load_chains.py [--heads] [--sites] [file.hip ...]        (default: every kernel file of pointcloudmatters_amd/csrc)

--heads: instead, per kernel, the number of times its head waits for outstanding SCALAR loads (s_load ... s_waitcnt lgkmcnt(0)) before the
         first vector load is issued (>= 2 listed): dependent scalar round trips at the start of every launch -- kernel arguments the
         compiler sank to their first use, a seed / an offset table read through a pointer, a job table searched by a scalar loop.
--sites: instead, per kernel, the number of vector loads that are followed by a full `s_waitcnt vmcnt(0)` before the next load is issued
         (>= 6 listed): load-and-wait inside per-element conditions (the attention kernels' `mask[key]`, FPS's per-point prologue)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-Xclang", "-target-feature",
         "-Xclang", "-packed-fp32-ops", "--cuda-device-only", "-S"]


def loops(lines):
    """(header index, back-branch index) of innermost loops: a label marked 'Inner Loop Header' up to the last branch back to it."""
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if m:
            lab = m.group(1)
            back = [j for j in range(i + 1, min(len(lines), i + 4000)) if re.search(r"s_cbranch\w*\s+%s\b" % re.escape(lab), lines[j])]
            if back:
                out.append((i, back[-1]))
    return out


def audit(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", asm], cwd=os.path.dirname(path), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    funcs = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    rows = []
    found = loops(lines)

    def nloads(a, b):
        return sum("global_load" in l or "buffer_load" in l for l in lines[a:b + 1])

    for a, b in found:
        body = lines[a:b + 1]
        loads = nloads(a, b)
        full = sum(bool(re.search(r"s_waitcnt vmcnt\(0\)", l)) for l in body)
        partial = sum(bool(re.search(r"s_waitcnt vmcnt\([1-9]", l)) for l in body)
        if loads and full and loads <= 2 and not partial:
            fn = [n for i, n in funcs if i < a][-1]
            lo = [i for i, n in funcs if i < a][-1]
            hi = min([i for i, n in funcs if i > a] + [len(lines)])
            if any(lo < a2 < hi and nloads(a2, b2) >= 4 for a2, b2 in found):
                continue  # the remainder loop of a kernel whose main loop keeps >= 4 loads in flight
            demangled = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()
            rows.append((demangled.replace("(anonymous namespace)::", "")[:70], loads, len(body)))
    return rows


def per_kernel(path, fn):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", asm], cwd=os.path.dirname(path), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(lines)
        ins = [l.split(";")[0].strip() for l in lines[i:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        name = subprocess.run(["c++filt", lines[i].split(":")[0]], capture_output=True, text=True).stdout.strip()
        yield name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80], fn(ins)


def head_waits(ins):
    waits, pending = 0, False
    for t in ins:
        if t.startswith("s_load_"):
            pending = True
        if re.match(r"s_waitcnt.*lgkmcnt\(0\)", t) and pending:
            waits, pending = waits + 1, False
        if t.startswith(("global_load", "buffer_load")):
            break
    return waits


def wait_sites(ins):
    n = 0
    for k, t in enumerate(ins):
        if t.startswith("global_load"):
            for t2 in ins[k + 1:k + 6]:
                if t2.startswith("global_load"):
                    break
                if re.match(r"s_waitcnt vmcnt\(0\)", t2):
                    n += 1
                    break
    return n


mode = "heads" if "--heads" in sys.argv else ("sites" if "--sites" in sys.argv else "loops")
sys.argv = [a for a in sys.argv if a not in ("--heads", "--sites")]
files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and f != "graph_fix.hip")
for f in files:
    if mode != "loops":
        fn, least, what = (head_waits, 2, "scalar round trips before the first vector load") if mode == "heads" else (wait_sites, 6, "load-then-full-wait sites")
        for name, v in per_kernel(os.path.abspath(f), fn):
            if v >= least:
                print("%-16s %3d %s   %s" % (os.path.basename(f), v, what, name))
        continue
    for name, loads, n in audit(os.path.abspath(f)):
        print("%-16s %-72s loads in flight %d, loop body %d lines" % (os.path.basename(f), name, loads, n))
