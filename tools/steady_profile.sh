#!/bin/bash
# usage: tools/steady_profile.sh <workload> [extra bench args]; prints the steady-state per-step kernel table
# (difference of a 30-step and a 10-step rocprofv3 run, which removes warm-up / MIOpen-find / capture kernels)
W=${1:-C2}; shift
cd /tmp && export TMPDIR=/tmp
for n in 10 30; do
  rm -rf /tmp/p$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$n -- python /root/repo/bench.py --workload $W --no-cpu-baseline --no-roofline --steps $n --warmup 5 "$@" > /tmp/b$n.log 2>&1
done
grep -o '"ms_per_step": [0-9.]*' /tmp/b30.log
python - <<PY
import csv,glob,re
def load(d):
    f=glob.glob(d+"/*/*kernel_stats.csv")[0]
    return {r["Name"]:(int(r["Calls"]),float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a,b=load("/tmp/p10"),load("/tmp/p30")
rows=[]
for k,(c,t) in b.items():
    c0,t0=a.get(k,(0,0.0)); rows.append(((t-t0)/20e3,(c-c0)/20,k))
rows.sort(reverse=True)
print("steady-state kernel time per step: %.3f ms, %d launches"%(sum(r[0] for r in rows)/1e3, sum(r[1] for r in rows)))
for us,c,k in rows[:${TOP:-45}]:
    k=re.sub(r"void |at::native::|\(anonymous namespace\)::","",k); m=re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)",k)
    print("%8.1f us %6.1f calls %7.2f avg  %s"%(us,c,us/max(c,1e-9),(m.group(1) if m else k[:90])))
PY
