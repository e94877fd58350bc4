"""Around every RCCL kernel of one optimizer step in a rocprofv3 kernel trace (rocpd sqlite): idle time before it (previous
kernel's end -> its start), its duration, idle time behind it (its end -> next kernel's start).
python tools/dbg/collective_gaps.py DIR/NAME_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name,start,end,stream_id from kernels order by start"))
ad = [i for i, r in enumerate(rows) if "adamw" in r[0] and (r[2] - r[1]) > 50e3]
step = rows[ad[-2] + 1:ad[-1] + 1]
print("launches", len(step), "span us", round((step[-1][2] - step[0][1]) / 1e3), "kernel time us", round(sum(r[2] - r[1] for r in step) / 1e3))
tot_b = tot_d = tot_a = 0
n = 0
for i, r in enumerate(step):
    if "ccl" in r[0].lower():
        before = (r[1] - max(x[2] for x in step[:i])) / 1e3 if i else 0.0
        after = (min((x[1] for x in step[i + 1:] if x[1] >= r[2]), default=r[2]) - r[2]) / 1e3
        dur = (r[2] - r[1]) / 1e3
        tot_b, tot_d, tot_a, n = tot_b + max(before, 0), tot_d + dur, tot_a + max(after, 0), n + 1
        print(f"  at {(r[1] - step[0][1]) / 1e3:8.1f} us  idle before {before:6.1f}  kernel {dur:6.1f}  idle after {after:6.1f}   {r[0][:60]}")
print(f"{n} collectives: idle before {tot_b:.0f} us, kernels {tot_d:.0f} us, idle after {tot_a:.0f} us")
# all idle gaps > 8 us in the step
ev = sorted((r[1], r[2]) for r in step)
ce = ev[0][1]
gaps = []
for s_, e_ in ev[1:]:
    if s_ > ce:
        gaps.append(((s_ - ce) / 1e3, (ce - step[0][1]) / 1e3))
    ce = max(ce, e_)
print("idle total us", round(sum(g for g, _ in gaps)), "gaps > 8 us:", [(round(g), round(t)) for g, t in gaps if g > 8][:60])
