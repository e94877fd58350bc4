#!/bin/bash
# Two ranks sharing ONE GPU (gloo carries the collectives; PCM_BENCH_SHARE_GPU test hook): round-3 hybrid mode against the
# round-4 segmented graph mode, same collectives, same contention.  Run on the GPU box.
for MODE in hybrid graph; do
  for WL in C2 C4; do
    PCM_BENCH_SHARE_GPU=1 PCM_DP_MODE=$MODE OMP_NUM_THREADS=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 40 --warmup 10 --workload $WL \
      --no-roofline --no-extra --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$MODE $WL', o['ms_per_step'], 'ms/step', o['value'], 'samples/s', o['config']['step_mode'], o['config'].get('gradient_exchange_exposed_ms'))"
  done
done
python bench.py --steps 40 --warmup 10 --no-roofline --no-extra --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('single C2', o['ms_per_step'], 'ms/step', o['config']['step_mode'])"
