"""One optimizer step of a rocprofv3 run (rocpd sqlite: `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...`)
as a kernel table with grids: python tools/dbg/step_kernels.py DIR/NAME_results.db [top] [--odd]
--odd: only the suspicious rows (framework kernels > 14 us, single-workgroup kernels > 12 us, own kernels > 60 us)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 50
odd = "--odd" in sys.argv
rows = list(db.cursor().execute("select name,start,end,grid_x,grid_y,workgroup_x,stream_id from kernels order by start"))
ad = [i for i, r in enumerate(rows) if "adamw" in r[0] and (r[2] - r[1]) > 50e3]
step = rows[ad[-2] + 1:ad[-1] + 1]
print("launches", len(step), "kernel time us", round(sum(r[2] - r[1] for r in step) / 1e3), "span us", round((step[-1][2] - step[0][1]) / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = r[0].replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")[:100] + f" g{r[3] // r[5]}x{r[4]} st{r[6]}"
    agg[k][0] += 1
    agg[k][1] += (r[2] - r[1]) / 1e3
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:top if not odd else 10000]:
    avg = v[1] / v[0]
    small = " g1x1 " in k or " g2x1 " in k
    if odd and not (("Cijk" not in k and "pcm_" not in k and avg > 14) or (small and avg > 12) or ("pcm_" in k and avg > 60)):
        continue
    print(f"{v[0]:4d} {avg:8.1f}us tot {v[1]:8.1f}  {k}")
