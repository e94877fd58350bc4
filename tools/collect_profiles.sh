#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Writes rocprofv3 outputs under
# gpurun_out/profiles_raw/<tag>/ ; tools/summarize_profiles.py turns them into profiles/*.{md,csv,json}.
#   pass A  kernel trace + stats of the default bench step (graph mode: every replayed kernel is visible)
#   pass B  kernel trace + stats of the per-kernel leg at C2 and at the HBM-sized shapes C3 / C5 / REF
#   pass C/D  PMC counters, ONE counter per pass, kernel-trace only (no sys/hip/hsa trace domains), per shape
set -x
TAG=${1:-r02}
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
find $OUT -name "*.csv" | head -40
