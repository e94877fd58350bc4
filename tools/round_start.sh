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
timeout 900 python bench.py --tables-out "$OUT/bench_tables.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cut -c1-400 "$OUT/bench.json"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 300 python tools/mb/mb_proj_ln.py > "$OUT/mb_proj_ln.log" 2>&1
echo "mb_proj_ln rc=$?"; cat "$OUT/mb_proj_ln.log" | tail -4
# the same bench with the MFMA projection chain on (opt-in until this number exists)
PCM_PROJ_MFMA=1 PCM_LINEAR_MFMA=1 PCM_PROJ_MFMA_LONG=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra > "$OUT/bench_mfma_chain.json" 2> "$OUT/bench_mfma_chain.err"
echo "bench (mfma chain) rc=$?"; cut -c1-200 "$OUT/bench_mfma_chain.json"
timeout 600 python bench.py --tokenizer-bf16 --no-cpu-baseline --no-roofline --no-extra > "$OUT/bench_tokenizer_bf16.json" 2> "$OUT/bench_tokenizer_bf16.err"
echo "bench (tokenizer bf16, the round-4 recipe) rc=$?"; cut -c1-200 "$OUT/bench_tokenizer_bf16.json"
cp -f "$GRAFT_REPO_ROOT/gpurun_out/pk_hazard.log" "$OUT/pk_hazard.log" 2>/dev/null  # written by tests/test_pk_hazard_gpu.py: copy to profiles/rNN_pk_hazard.log
# the rocprofv3 kernel trace of the default bench step (pass A of tools/collect_profiles.sh): the per-kernel table profiles/rNN_summary.md is
# built from (tools/summarize_profiles.py) -- in this call already, in case it is the only one the round gets
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "gpurun_out/profiles_raw/$TAG/bench" -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra \
    > "$OUT/rocprof_bench.log" 2>&1 )
echo "rocprofv3 bench rc=$?"; ls gpurun_out/profiles_raw/$TAG/bench 2>/dev/null | head -5
