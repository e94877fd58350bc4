"""Ordered kernel list of one optimizer step from a rocprofv3 rocpd database (see step_kernels.py):
python tools/dbg/step_sequence.py DB > sequence.txt   -- one line per launch: index, start (us), duration, grid, stream, name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name,start,end,grid_x,grid_y,workgroup_x,stream_id from kernels order by start"))
ad = [i for i, r in enumerate(rows) if "adamw" in r[0] and (r[2] - r[1]) > 50e3]
step = rows[ad[-2] + 1:ad[-1] + 1]
t0 = step[0][1]
for i, r in enumerate(step):
    name = r[0].replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    print(f"{i:4d} {(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} g{r[3] // r[5]}x{r[4]} s{r[6]} {name[:110]}")
