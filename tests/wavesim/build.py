"""Builds tests/wavesim/_build/libpcm_wavesim.so: the product's kernel sources (pointcloudmatters_amd/csrc/*.hip) compiled for the HOST
against the wave64 execution model of wavesim.hpp, exporting the same extern "C" entry points as libpcm_pointops.so -- with host
pointers in place of device pointers.  TESTS ONLY (tests/test_wavesim_parity.py); nothing in the product can load it.

The sources are compiled as they are, except for these mechanical rewrites of constructs a host compiler cannot take:
  1. `extern __shared__ <type> name[];`            -> `<type> *name = (<type> *)wavesim::dyn_smem();`   (dynamic LDS)
  2. fps.hip `fps_wave_max_fast`: the inline-assembly DPP ladder (6 x v_max_u32_dpp + readlane) -> `pcm_wave_max_u32`, the same
     wave maximum written with the DPP builtin in pcm_common.hpp (the model executes DPP controls, not assembly text)
  3. `asm volatile("" ...)` statements -> removed: scheduling fences (ffn.hip) and the "Kernel heads" statements of pcm_common.hpp (empty asm
     that names kernel arguments as register inputs, or makes the dropout seed / a mask byte opaque at its use): they emit no instruction and
     change no value, only where the device compiler places loads and waits
  4. knn.hip `pcm_knn_exact_kernel` of the FROZEN file only (csrc/knn.hip, not csrc/next/knn.hip): the 64 lanes initialise the LDS heap and
     all of them read its root in the next statement.  On hardware a wave's LDS operations execute in program order for all its lanes
     at once; the model runs lanes one after another and needs a `__builtin_amdgcn_wave_barrier()` behind the initialisation loop.
     Everywhere else such hand-offs carry that statement IN THE SOURCE (attn_small.hip `tile_store`, the heap sort of this kernel, both
     places in next/knn.hip): it emits no instruction.  Here it cannot: with it the compiler schedules the kernel differently, and the
     frozen file's gfx950 code must stay byte-identical to the hardware-tested build (tests/test_build_flags.py).
Compiler: the ROCm clang++ in host mode (ext_vector_type, __builtin_convertvector on __bf16), -ffp-contract=off like the device build.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")
# WAVESIM_SANITIZE=1: the same library with AddressSanitizer + UndefinedBehaviorSanitizer instrumentation of the KERNEL SOURCES (own
# directory; the process needs the clang runtime preloaded: asan_runtime()).  Device-side sanitizers cannot run on this pool (no xnack);
# on the model an access one element outside a global buffer, an LDS array or the dynamic LDS block aborts the test.
SANITIZE = os.environ.get("WAVESIM_SANITIZE") == "1"
# WAVESIM_VARIANT=next: the ten files of csrc/ that are frozen at the last hardware-tested sources are taken from csrc/next/ instead (round 5's
# ISA rewrites, `make next`; csrc/Makefile) -- the same curated tests then run against the rewrites.  Default: what ships.
VARIANT = os.environ.get("WAVESIM_VARIANT", "")
assert VARIANT in ("", "next"), VARIANT
NEXT_FILES = ("fps", "knn", "drln", "ffn", "attn_small", "attn_flash", "tokens", "optim", "sa_fused", "bnrelu")
OUT = os.path.join(HERE, "_build", *([VARIANT] if VARIANT else []), *(["asan"] if SANITIZE else []))
LIB = os.path.join(OUT, "libpcm_wavesim.so")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function,alignment", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
             "-shared-libsan"] if SANITIZE else []


def asan_runtime():
    import glob

    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None

CLANG = os.environ.get("WAVESIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
# every kernel file of the library except graph_fix.hip (hipGraph surgery: runtime API, no kernel logic)
SOURCES = ["fps.hip", "knn.hip", "ball.hip", "group.hip", "misc_ops.hip", "segsum.hip", "voxel.hip", "sa_scatter.hip", "sa_fused.hip",
           "bnrelu.hip", "drln.hip", "tokens.hip", "gnmish.hip", "ddpm.hip", "optim.hip", "ffn.hip", "ffn_mfma.hip", "attn_small.hip",
           "attn_flash.hip", "proj_ln.hip"] + ([] if VARIANT == "next" else ["bnact.hip"])  # next/bnrelu.hip exports the pcm_bn_act_* entry points itself


def source_path(name):
    if VARIANT == "next" and name[:-4] in NEXT_FILES:
        return os.path.join(CSRC, "next", name)
    return os.path.join(CSRC, name)
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-g0" if not SANITIZE else "-g1", "-ffp-contract=off", "-fPIC", "-fno-strict-aliasing", "-Wno-everything",
         "-I", os.path.join(HERE, "include"), "-I", CSRC] + SAN_FLAGS

_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s*\b([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*\]\s*;")
_FPS_ASM = re.compile(r"(__device__ __forceinline__ uint32_t fps_wave_max_fast\(uint32_t v\)\s*\{).*?\n\}\n", re.S)
_FENCE = re.compile(r"asm volatile\(\"\"[^;]*\);")


def rewrite(name, text):
    text = _DYN.sub(lambda m: f"{m.group(1)} *{m.group(2)} = ({m.group(1)} *)wavesim::dyn_smem();", text)
    if name == "fps.hip":
        text, n = _FPS_ASM.subn(r"\1\n    return pcm_wave_max_u32(v);  // tests/wavesim/build.py, rewrite 2\n}\n", text)
        assert n == 1, "fps_wave_max_fast not found"
    if name == "knn.hip" and VARIANT != "next":
        wb = " __builtin_amdgcn_wave_barrier();  /* tests/wavesim/build.py, rewrite 4 */"
        text, n1 = re.subn(r"(for \(int i = lane; i < nsample; i \+= 64\) bd\[i\] = 1e10f, bi\[i\] = -1;)", r"\1" + wb, text)
        assert n1 == 1, n1
    text = _FENCE.sub("/* scheduling fence removed (tests/wavesim/build.py, rewrite 3) */;", text)
    assert "asm" not in re.sub(r"//.*", "", text).replace("assume", ""), f"{name}: inline assembly left after the rewrites"
    return text


def build(verbose=False, sources=None):
    """Idempotent and safe to call from several processes at once (pytest-xdist workers): serialised by a lock file."""
    import fcntl

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(verbose, sources)


def _build(verbose, sources):
    if not os.path.exists(CLANG):
        raise RuntimeError(f"{CLANG} not found (set WAVESIM_CXX)")
    objs, jobs = [], []
    deps = [os.path.join(HERE, f) for f in ("wavesim.hpp", "wavesim.cpp", "include/hip/hip_runtime.h", "include/hip/hip_bf16.h", "build.py")] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "pcm_pointops.h")]
    stamp = hashlib.sha1(b"".join(open(d, "rb").read() for d in sorted(deps)) + (b"asan" if SANITIZE else b"")).hexdigest()[:12]
    for name in (sources or SOURCES):
        src = rewrite(name, open(source_path(name)).read())
        key = hashlib.sha1((stamp + src).encode()).hexdigest()[:16]
        cpp, obj = os.path.join(OUT, name.replace(".hip", ".sim.cpp")), os.path.join(OUT, name.replace(".hip", f".{key}.o"))
        objs.append(obj)
        if not os.path.exists(obj):
            open(cpp, "w").write(src)
            jobs.append((name, subprocess.Popen([CLANG] + FLAGS + ["-c", cpp, "-o", obj], stderr=subprocess.PIPE, text=True)))
    core = os.path.join(OUT, f"wavesim.{stamp}.o")
    if not os.path.exists(core):
        jobs.append(("wavesim.cpp", subprocess.Popen([CLANG, "-std=c++17", "-O2", "-fPIC"] + SAN_FLAGS + ["-c", os.path.join(HERE, "wavesim.cpp"), "-o", core],
                                                     stderr=subprocess.PIPE, text=True)))
    failed = []
    for name, p in jobs:
        err = p.communicate()[1]
        if p.returncode != 0:
            failed.append((name, err))
        elif verbose and err.strip():
            print(err, file=sys.stderr)
    if failed:
        raise RuntimeError("\n".join(f"--- {n}\n{e[-6000:]}" for n, e in failed))
    if jobs or not os.path.exists(LIB):
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.check_call([CLANG, "-shared", "-fPIC"] + SAN_FLAGS + ["-o", tmp] + objs + [core])
        os.replace(tmp, LIB)  # processes that already mapped the previous file keep it
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, sources=sys.argv[1:] or None))
