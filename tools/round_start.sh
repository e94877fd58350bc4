#!/bin/bash
# Run ON THE GPU BOX as the FIRST gpurun call of a round:   gpurun --timeout 1800 -- 'bash tools/round_start.sh r05'
# Round 4 ended with the GPU pool closed for its whole last session: HEAD's host-side changes of that session (status-contract sweep,
# device code byte-identical: profiles/r04_device_digest.txt) and the pending RowsLinear patch have not met the hardware yet.
#   1  full -m gpu suite at HEAD                                   -> gpurun_out/<tag>/gpu_tests.log
#   2  default bench line                                          -> gpurun_out/<tag>/bench.json
#   3  the pending patch on a scratch copy, its targeted tests     -> gpurun_out/<tag>/pending_rowslinear.log
#      (green there = `git apply tools/dbg/pending_rowslinear.patch` in the build container and commit)
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/gpu_tests.log" 2>&1
echo "suite rc=$?" | tee -a "$OUT/gpu_tests.log"
tail -3 "$OUT/gpu_tests.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cut -c1-400 "$OUT/bench.json"
SCR=/tmp/pcm_pending
rm -rf $SCR && mkdir -p $SCR && cp -r . $SCR/ 2>/dev/null
cd $SCR && rm -rf gpurun_out && patch -p1 -s < tools/dbg/pending_rowslinear.patch \
  && timeout 900 python -m pytest -q -p no:cacheprovider tests/test_wrappers_ref_gpu.py tests/test_rows_linear_gpu.py tests/test_unet_ops_gpu.py tests/test_wide_fixture.py \
       tests/test_concurrency_gpu.py tests/test_policy_gpu.py tests/test_pointnet2_gpu.py tests/test_rlbench_gpu.py tests/test_presample.py \
       > "$OUT/pending_rowslinear.log" 2>&1
echo "pending patch rc=$?" | tee -a "$OUT/pending_rowslinear.log"
tail -3 "$OUT/pending_rowslinear.log"
