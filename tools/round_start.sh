#!/bin/bash
# Run ON THE GPU BOX as the FIRST gpurun call of a round:   gpurun --timeout 2400 -- 'bash tools/round_start.sh r05'
# Rounds 4 and 5 began with the GPU pool closed from outside the build: whatever was committed meanwhile has not met the hardware.
#   1  full -m gpu suite at HEAD (no -x: every failure is wanted)   -> gpurun_out/<tag>/gpu_tests.log
#   2  default bench line                                            -> gpurun_out/<tag>/bench.json (+ bench_tables.json)
#   3  smoke()                                                       -> gpurun_out/<tag>/smoke.log
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests.log" 2>&1
echo "suite rc=$?" | tee -a "$OUT/gpu_tests.log"
tail -3 "$OUT/gpu_tests.log"
timeout 600 python bench.py --tables-out "$OUT/bench_tables.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cut -c1-400 "$OUT/bench.json"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
