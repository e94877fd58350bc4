#!/bin/bash
# PMC traffic passes only (FETCH_SIZE / WRITE_SIZE, one counter per pass, kernel-trace only) + kernel stats at the HBM-sized shapes.
# Run ON THE GPU BOX from the repo root; same layout as tools/collect_profiles.sh so tools/summarize_profiles.py reads it.
TAG=${1:-r04}
OUT=gpurun_out/profiles_raw/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in C3 C5 REF; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernels_$S -o kernels -- python tools/pmc_workload.py $S > $OUT/kernels_$S.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$S -o pmc -- python tools/pmc_workload.py $S > $OUT/pmc_fetch_$S.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$S -o pmc -- python tools/pmc_workload.py $S > $OUT/pmc_write_$S.log 2>&1
done
