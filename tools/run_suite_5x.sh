#!/bin/bash
# the GPU suite five times without -x and once with -x (the driver's command), keeping the result lines.
# usage (on the GPU box): tools/run_suite_5x.sh [round-tag]   -> gpurun_out/<tag>_gpu_tests_run{1..5,x}.log
tag=${1:-r03}
mkdir -p gpurun_out
git_head=$(cat .git_head 2>/dev/null || echo unknown)
for i in 1 2 3 4 5 x; do
  flags="-q"; [ "$i" = x ] && flags="-x -q"
  out=gpurun_out/${tag}_gpu_tests_run${i}.log
  echo "# python -m pytest tests/ $flags -m gpu   (HEAD $git_head, run $i)" > $out
  python -m pytest tests/ $flags -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | grep -v "NCCL\|RCCL" >> $out
  tail -1 $out
done
