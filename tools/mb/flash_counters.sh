#!/bin/bash
# SQ counter passes (one counter per pass, kernel-trace only) over the flash attention kernels at one shape.
# usage (GPU box): tools/mb/flash_counters.sh 2051 [p_drop]  -> gpurun_out/flash_ctr/<counter>.csv summary on stdout
S=${1:-2051}; P=${2:-0.1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/flash_ctr; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/avail.txt
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_TRANS SQ_WAIT_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM; do
  grep -qx $C $OUT/avail.txt || { echo "$C not available"; continue; }
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o pmc -- python tools/mb/mb_attn_short.py $P ${S}x${S} > $OUT/$C.log 2>&1 || echo "$C failed"
done
python - <<PY
import csv,glob,collections,os
out=collections.defaultdict(dict)
for f in glob.glob("$OUT/*/*counter_collection.csv")+glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "flash" not in k: continue
        k=k.split("(")[0].replace("(anonymous namespace)::","").split("pcm_attn_")[-1]
        c=r["Counter_Name"]; v=float(r["Counter_Value"])
        d=out[k].setdefault(c,[0,0.0]); d[0]+=1; d[1]+=v
for k,v in out.items():
    print(k)
    for c,(n,t) in sorted(v.items()): print(f"   {c:32s} {t/n:14.1f}  (n={n})")
PY
