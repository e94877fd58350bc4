"""Workload for the rocprofv3 --pmc passes: a calibration copy with a KNOWN byte count (so that the
FETCH_SIZE / WRITE_SIZE -> bytes factors are measured on this box, in a 16 B/lane streaming pattern,
as /opt/skills/guides/MI355X_MICROARCH.md section HBM asks) followed by the per-kernel leg of bench.py at
one shape: C2 (the workload, all kernels) or C3 / C5 / REF (the gather / scatter / sampling kernels at HBM-sized shapes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from pointcloudmatters_amd.bc import WORKLOADS  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda:0")
# calibration: 512 MiB fp32 copy = 512 MiB read + 512 MiB written, larger than the 256 MiB Infinity Cache
src = torch.randn(128 * 1024 * 1024, device=dev)
dst = torch.empty_like(src)
for _ in range(5):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst
print(json.dumps({"calibration": {"kernel": "copy", "bytes_read": 128 * 1024 * 1024 * 4, "bytes_written": 128 * 1024 * 1024 * 4}}))
if shape in bench.HBM_SHAPES:
    print(json.dumps({"kernels_hbm": bench.kernel_rooflines_hbm(dev, [shape])}))
else:
    print(json.dumps({"kernels": bench.kernel_rooflines(WORKLOADS[shape], dev)}))
