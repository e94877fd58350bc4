#!/bin/bash
# SQ counter passes (one counter per pass, kernel-trace only) for the kernels matching a name pattern in any command.
# usage (GPU box): tools/mb/kernel_counters.sh <kernel-regex> <out-tag> -- <command ...>
PAT=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ctr_$TAG; mkdir -p $OUT
for C in SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o pmc -- "$@" > $OUT/$C.log 2>&1 || echo "$C failed"
done
python - <<PY
import csv,glob,collections,re
out=collections.defaultdict(dict)
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not re.search(r"$PAT", k): continue
        k=re.sub(r"\(.*","",k.replace("(anonymous namespace)::","").replace("void ",""))+" g"+r["Grid_Size"]
        d=out[k].setdefault(r["Counter_Name"],[0,0.0]); d[0]+=1; d[1]+=float(r["Counter_Value"])
ks=sorted(out)
cs=sorted({c for k in ks for c in out[k]})
for k in ks:
    print(k)
    for c in cs:
        if c in out[k]: print(f"   {c:26s} {out[k][c][1]/out[k][c][0]:16.0f}")
PY
