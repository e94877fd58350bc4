"""Which KERNELS of a changed .hip file have a different gfx950 instruction stream than at a git revision?

    python tools/device_kernel_diff.py <rev> ffn sa_fused tokens ...

Compiles both versions of each file to assembly with the library's flags (the revision's headers beside its sources), splits the text by
kernel, strips labels / comments / directives and compares.  tools/device_code_digest.sh answers "did the object change"; this answers
"which kernels" -- e.g. that a one-kernel edit left the other 24 kernels of csrc/sa_fused.hip instruction for instruction as they were
when the hardware tests last ran (profiles/r05_device_kernel_diff.txt)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-Xclang", "-target-feature",
         "-Xclang", "-packed-fp32-ops", "--cuda-device-only", "-S"]


def kernels(asm):
    lines = open(asm).read().splitlines()
    idx = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    out = {}
    for i, n in idx:
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        body = [re.sub(r"\.LBB\d+_\d+|;.*$", "", l).strip() for l in lines[i + 1:end + 1]]
        out[n] = [b for b in body if b and not b.startswith(".")]
    return out


def compile_to(src_dir, name, out):
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [name + ".hip", "-o", out], cwd=src_dir, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)


rev, names = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    old = os.path.join(td, "pointcloudmatters_amd", "csrc")
    os.makedirs(old)
    os.makedirs(os.path.join(td, "include"))
    listing = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "pointcloudmatters_amd/csrc/", "include/"], capture_output=True,
                             text=True, check=True).stdout.split()
    for path in listing:
        if path.endswith((".hip", ".hpp", ".h")):
            blob = subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, path)], capture_output=True, check=True).stdout
            open(os.path.join(td, path), "wb").write(blob)
    for n in names:
        a_s, b_s = os.path.join(td, n + ".old.s"), os.path.join(td, n + ".new.s")
        compile_to(old, n, a_s)
        compile_to(CSRC, n, b_s)
        a, b = kernels(a_s), kernels(b_s)
        same = [k for k in b if k in a and a[k] == b[k]]
        diff = [k for k in b if k not in a or a[k] != b[k]]
        gone = [k for k in a if k not in b]
        dm = lambda ks: ", ".join(sorted({subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                                          .replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for k in ks}))
        print("%-12s %2d kernels, %2d with the instruction stream of %s; changed / new: %s%s" % (
            n + ".hip", len(b), len(same), rev, dm(diff) or "-", ("; removed: " + dm(gone)) if gone else ""))
