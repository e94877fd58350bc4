#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Writes rocprofv3 outputs under
# gpurun_out/profiles_raw/<tag>/ ; tools/summarize_profiles.py turns them into profiles/*.{md,csv,json}.
#   pass A  kernel trace + stats of the default bench step (graph mode: every replayed kernel is visible)
#   pass B  kernel trace + stats of the per-kernel leg at C2 and at the HBM-sized shapes C3 / C5 / REF
#   pass C/D  PMC counters, ONE counter per pass, kernel-trace only (no sys/hip/hsa trace domains), per shape
set -x
TAG=${1:-r03}
OUT=gpurun_out/profiles_raw/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $OUT/bench.log 2>&1
for S in C2 C3 C5 REF; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernels_$S -o kernels -- \
      python tools/pmc_workload.py $S > $OUT/kernels_$S.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$S -o pmc -- \
      python tools/pmc_workload.py $S > $OUT/pmc_fetch_$S.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$S -o pmc -- \
      python tools/pmc_workload.py $S > $OUT/pmc_write_$S.log 2>&1
done
# pass E  SQ counters (instruction mix, LDS waits, MFMA busy), ONE counter per pass, kernel-trace only, at C2 and REF: what
#         backs "latency-bound" / "VALU-bound" / "LDS-bound" in DESIGN.md with numbers
for S in C2 REF; do
  for C in SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/sq_${C}_$S -o pmc -- \
        python tools/pmc_workload.py $S > $OUT/sq_${C}_$S.log 2>&1 || echo "counter $C failed at $S"
  done
done
find $OUT -name "*.csv" | head -60
