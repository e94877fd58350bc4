"""rocprofv3 --kernel-trace CSV -> launches in order, grouped by (kernel, grid): count, mean us, min us.  Optional substring filter.
python tools/dbg/trace_by_kernel.py DIR [substr]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if sub not in name:
        continue
    key = (name[:48], r.get("Grid_Size_X", r.get("Grid_Size", "")))
    out.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in out.items():
    print(f"{k[0]:50s} grid {k[1]:>8s}  n={len(v):4d}  mean {sum(v) / len(v):8.1f} us  min {min(v):8.1f} us")
