#!/usr/bin/env python
"""Print a compact view of a rocprofv3 kernel_stats.csv: calls, total ms, average us, short kernel name."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0  # e.g. number of timed calls, to print per-call figures
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / div:.3f} ms" + (f" per unit (/{div:g})" if div != 1 else ""))
for r in rows[:top]:
    n = r["Name"]
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", n)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", n)
    short = m.group(1) if m else n[:110]
    print(f"{int(r['Calls']) / div:9.1f} {float(r['TotalDurationNs']) / 1e6 / div:9.3f} ms {float(r['AverageNs']) / 1e3:9.2f} us  {short}")
