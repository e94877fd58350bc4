"""The device code of the product library contains no packed-fp32 VALU instructions (csrc/Makefile NO_PK; why: tests/test_concurrency_gpu.py).
Compiles two kernel files to assembly with the Makefile's own flags (hipcc cross-compiles without a GPU) and greps."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloudmatters_amd", "csrc")


def _flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    no_pk = re.search(r"^NO_PK\s*\?=\s*(.+)$", mk, re.M).group(1).split()
    flags = re.search(r"^FLAGS\s*\?=\s*(.+)$", mk, re.M).group(1)
    assert "$(NO_PK)" in flags, "FLAGS must carry $(NO_PK)"
    asan = re.search(r"^ASAN_FLAGS\s*:=\s*(.+)$", mk, re.M).group(1)
    assert "$(NO_PK)" in asan
    out = []
    for tok in flags.split():
        out += no_pk if tok == "$(NO_PK)" else [tok.replace("$(ARCH)", "gfx950")]
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["fps.hip", "drln.hip"])
def test_no_packed_fp32_instructions_in_device_code(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / (src + ".s")
    r = subprocess.run([hipcc] + _flags() + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert "amdgcn" in asm and "gfx950" in asm
    hits = re.findall(r"v_pk_(?:add|mul|fma)_f32", asm)
    assert not hits, "%d packed-fp32 instructions in %s" % (len(hits), src)
