#!/bin/bash
# Run ON THE GPU BOX as the FIRST gpurun call of a round:   gpurun --timeout 4500 -- 'bash tools/round_start.sh r06'
# Rounds 4 - 6 had the GPU pool closed from outside the build; round 5 rewrote the device code of ten files without a hardware run.
# The shipped library (lib/) carries the round-4 device code of those files (byte-identical to the last hardware-tested build,
# tests/test_build_flags.py); the rewrites build into lib_next/ (`make -C pointcloudmatters_amd/csrc next`).  Every leg runs on both:
#   1  full -m gpu suite on the shipped library (no -x: every failure is wanted) -> gpurun_out/<tag>/gpu_tests.log
#   2  the rewritten files' test tiers against lib_next               -> gpurun_out/<tag>/gpu_tests_next.log
#   3  bench line (library products), shipped and lib_next            -> gpurun_out/<tag>/bench.json, bench_next.json
#   4  smoke()                                                        -> gpurun_out/<tag>/smoke.log
#   5  the MFMA projection chain forced on / the bf16 tokenizer       -> bench_mfma_*.json, bench_tokenizer_bf16.json
#   6  rocprofv3 --kernel-trace --stats of the bench step, both libs  -> gpurun_out/profiles_raw/<tag>/{bench,bench_next}
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
BASE=$GRAFT_REPO_ROOT/pointcloudmatters_amd/lib_next/libpcm_pointops.so
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT" || exit 1
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > "$OUT/box.txt"; nproc >> "$OUT/box.txt"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests.log" 2>&1
echo "suite rc=$?" | tee -a "$OUT/gpu_tests.log"
tail -3 "$OUT/gpu_tests.log"
if [ -f "$BASE" ]; then
  PCM_POINTOPS_LIB=$BASE timeout 900 python -m pytest tests/test_pointops_gpu.py tests/test_pointops_fuzz_gpu.py tests/test_drln_gpu.py \
      tests/test_small_attn_gpu.py tests/test_flash_attn_gpu.py tests/test_tokens_gpu.py tests/test_xfer_gpu.py tests/test_sa_fused_gpu.py tests/test_bn_relu_gpu.py \
      -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests_next.log" 2>&1
  echo "suite (lib_next, rewritten files' tests) rc=$?" | tee -a "$OUT/gpu_tests_next.log"; tail -2 "$OUT/gpu_tests_next.log"
  # next/fps.hip uses packed fp32 (plain forms only): its bit-exact tests BESIDE a busy device (a replayed GEMM graph on another stream), the
  # load under which round 4's hazard showed
  PCM_TEST_BUSY=1 PCM_POINTOPS_LIB=$BASE timeout 600 python -m pytest tests/test_pointops_gpu.py tests/test_pointops_fuzz_gpu.py -k "fps" \
      -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests_next_fps_busy.log" 2>&1
  echo "FPS tests on lib_next beside a busy device rc=$?"; tail -1 "$OUT/gpu_tests_next_fps_busy.log"
  # ... and the other files lib_next builds with packed fp32 (op_sel_hi / neg forms only): their kernel-level tests beside a busy device
  PCM_TEST_BUSY=1 PCM_POINTOPS_LIB=$BASE timeout 900 python -m pytest tests/test_small_attn_gpu.py tests/test_flash_attn_gpu.py tests/test_bn_relu_gpu.py \
      tests/test_tokens_gpu.py tests/test_pointops_gpu.py -k "not fps" -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests_next_pk_busy.log" 2>&1
  echo "packed-fp32 files of lib_next beside a busy device rc=$?"; tail -1 "$OUT/gpu_tests_next_pk_busy.log"
fi
timeout 900 python bench.py --tables-out "$OUT/bench_tables.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cut -c1-400 "$OUT/bench.json"
if [ -f "$BASE" ]; then
  PCM_POINTOPS_LIB=$BASE timeout 600 python bench.py --no-cpu-baseline --no-extra --tables-out "$OUT/bench_tables_next.json" > "$OUT/bench_next.json" 2> "$OUT/bench_next.err"
  echo "bench (lib_next) rc=$?"; cut -c1-300 "$OUT/bench_next.json"
fi
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 300 python tools/mb/mb_proj_ln.py > "$OUT/mb_proj_ln.log" 2>&1
echo "mb_proj_ln rc=$?"; cat "$OUT/mb_proj_ln.log" | tail -6
# the same bench with the MFMA projection chain forced on (opt-in until these numbers exist), short sites and short + long sites
PCM_PROJ_MFMA=1 PCM_LINEAR_MFMA=1 PCM_PROJ_MFMA_LONG=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --emit-warmup-losses > "$OUT/bench_mfma_short.json" 2> "$OUT/bench_mfma_short.err"
echo "bench (mfma chain, short sites) rc=$?"; cut -c1-200 "$OUT/bench_mfma_short.json"
PCM_PROJ_MFMA=1 PCM_LINEAR_MFMA=1 PCM_PROJ_MFMA_LONG=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --emit-warmup-losses > "$OUT/bench_mfma_long.json" 2> "$OUT/bench_mfma_long.err"
echo "bench (mfma chain, short + long sites) rc=$?"; cut -c1-200 "$OUT/bench_mfma_long.json"
PCM_PROJ_MFMA=1 PCM_LINEAR_MFMA=1 PCM_PROJ_MFMA_BWD=1 PCM_LINEAR_MFMA_BWD=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --emit-warmup-losses > "$OUT/bench_mfma_short_bwd.json" 2> "$OUT/bench_mfma_short_bwd.err"
echo "bench (mfma chain, short sites, forward + backward) rc=$?"; cut -c1-200 "$OUT/bench_mfma_short_bwd.json"
PCM_PROJ_MFMA=0 PCM_LINEAR_MFMA=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --emit-warmup-losses > "$OUT/bench_lib_losses.json" 2> "$OUT/bench_lib_losses.err"
echo "bench (library products, with warm-up losses) rc=$?"; cut -c1-200 "$OUT/bench_lib_losses.json"
timeout 600 python bench.py --tokenizer-bf16 --no-cpu-baseline --no-roofline --no-extra > "$OUT/bench_tokenizer_bf16.json" 2> "$OUT/bench_tokenizer_bf16.err"
echo "bench (tokenizer bf16, the round-4 recipe) rc=$?"; cut -c1-200 "$OUT/bench_tokenizer_bf16.json"
# the Diffusion-Policy workload, projector as modules (default) and in row layout (opt-in: csrc/bnact.hip's first contact)
timeout 600 python bench.py --workload C3 --no-cpu-baseline --no-roofline --no-extra > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
echo "bench C3 rc=$?"; cut -c1-160 "$OUT/bench_c3.json"
PCM_PROJECTOR_ROWS=1 timeout 600 python bench.py --workload C3 --no-cpu-baseline --no-roofline --no-extra > "$OUT/bench_c3_rows.json" 2> "$OUT/bench_c3_rows.err"
echo "bench C3 (row-layout projector) rc=$?"; cut -c1-160 "$OUT/bench_c3_rows.json"
cp -f "$GRAFT_REPO_ROOT/gpurun_out/pk_hazard.log" "$OUT/pk_hazard.log" 2>/dev/null  # written by tests/test_pk_hazard_gpu.py: copy to profiles/rNN_pk_hazard.log
# the rocprofv3 kernel trace of the bench step on both builds: the per-kernel old-vs-new table of profiles/rNN_summary.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/profiles_raw/$TAG/bench" -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$OUT/rocprof_bench.log" 2>&1
echo "rocprofv3 bench rc=$?"; ls gpurun_out/profiles_raw/$TAG/bench 2>/dev/null | head -5
if [ -f "$BASE" ]; then
  PCM_POINTOPS_LIB=$BASE timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/profiles_raw/$TAG/bench_next" -o bench -- \
      python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$OUT/rocprof_bench_next.log" 2>&1
  echo "rocprofv3 bench (lib_next) rc=$?"
fi
# keep the merged-back size small: only the stats / kernel-trace summaries travel (the 64 MiB limit)
find gpurun_out/profiles_raw/$TAG -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh gpurun_out 2>/dev/null
