#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Writes rocprofv3 outputs under
# gpurun_out/profiles_raw/ ; tools/summarize_profiles.py turns them into profiles/*.{md,csv,json}.
#   pass A  kernel trace + stats of the default bench (graph mode: every replayed kernel is visible)
#   pass B  kernel trace of the per-kernel leg (bench.py --kernels-only)
#   pass C/D  PMC counters, ONE counter group per pass, kernel-trace only (no sys/hip/hsa trace domains)
set -x
TAG=${1:-r01}
OUT=gpurun_out/profiles_raw/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernels -o kernels -- \
    python bench.py --kernels-only > $OUT/kernels.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- \
    python tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- \
    python tools/pmc_workload.py > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -50
