#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the bench lines that DESIGN.md / README.md quote, written to
# gpurun_out/bench_<tag>/ ; copy what is to be judged into profiles/ afterwards (tools/summarize_profiles.py does not).
TAG=${1:-r03}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
python bench.py --tables-out $OUT/bench_tables.json > $OUT/bench_output.json 2> $OUT/bench_output.err
: > $OUT/other_workloads.jsonl
for W in C2R C3 C3R C4 C4N C5 C5B RLB RLBDP REF; do
  python bench.py --workload $W --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | grep '"metric"' >> $OUT/other_workloads.jsonl
done
for D in prune_backward skip; do
  python bench.py --dead-decoder-layers $D --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | grep '"metric"' >> $OUT/other_workloads.jsonl
done
for M in hybrid flat eager; do
  python bench.py --mode $M --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | grep '"metric"' >> $OUT/other_workloads.jsonl
done
python bench.py --workload REF --no-prefetch --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | grep '"metric"' >> $OUT/other_workloads.jsonl
python tools/bench_rollout.py > $OUT/rollout_latency.jsonl 2>/dev/null
wc -l $OUT/*
