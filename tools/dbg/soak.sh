# Long runs of the replayed training step (graph / hybrid modes): thousands of replays, loss finite at the end, two runs agree.
# Run on the GPU box from the repo root: bash tools/dbg/soak.sh > gpurun_out/r04_soak.log
for W in C2 C3 REF C2P; do
  for rep in 1 2; do
    python bench.py --workload $W --steps 1500 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W run $rep:', d['steps'], 'steps', d['value'], d['unit'], d['ms_per_step'], 'ms/step, final loss', d['final_loss'], 'mode', d['config'].get('step_mode'))"
  done
done
