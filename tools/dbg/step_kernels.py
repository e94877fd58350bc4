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

if "--gaps" in sys.argv:  # idle time of the device inside the step (union over streams)
    ev = sorted((r[1], r[2]) for r in step)
    cs, ce = ev[0]
    busy, gaps = 0, []
    for s_, e_ in ev[1:]:
        if s_ > ce:
            gaps.append((s_ - ce, ce - step[0][1]))
            busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    print("union busy us", round(busy / 1e3), "idle us", round(sum(g for g, _ in gaps) / 1e3), "gaps > 15 us (us, at us):",
          [(round(g / 1e3), round(t / 1e3)) for g, t in gaps if g > 15e3][:40])
    per = collections.Counter(r[6] for r in step)
    for sid, n in per.items():
        ks = [r for r in step if r[6] == sid]
        print(" stream", sid, "kernels", n, "busy us", round(sum(r[2] - r[1] for r in ks) / 1e3), "from", round((ks[0][1] - step[0][1]) / 1e3), "to", round((ks[-1][2] - step[0][1]) / 1e3))
