#!/bin/bash
# The short form of tools/round_start.sh for a GPU call late in a session (~30 minutes): the -m gpu suite on the shipped library, the default
# bench line, smoke(), and the rocprofv3 kernel trace of the bench step.  Same output layout (gpurun_out/<tag>/, profiles_raw/<tag>/bench).
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT" || exit 1
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > "$OUT/box.txt"; nproc >> "$OUT/box.txt"
timeout 900 python bench.py --tables-out "$OUT/bench_tables.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/profiles_raw/$TAG/bench" -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$OUT/rocprof_bench.log" 2>&1
echo "rocprofv3 bench rc=$?"
find gpurun_out/profiles_raw/$TAG -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests.log" 2>&1
echo "suite rc=$?" | tee -a "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
