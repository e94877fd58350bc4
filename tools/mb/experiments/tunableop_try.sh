set -x
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFAULT', d['value'], d['ms_per_step'])"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_c2.csv PYTORCH_TUNABLEOP_VERBOSE=1
PYTORCH_TUNABLEOP_TUNING=1 timeout 2400 $B > gpurun_out/tune_run.json 2> gpurun_out/tune_run.err
tail -c 400 gpurun_out/tune_run.err
python -c "import sys,json; d=json.loads(open('gpurun_out/tune_run.json').read().strip().splitlines()[-1]); print('TUNING-RUN', d['value'], d['ms_per_step'])"
ls -la gpurun_out/tunableop_c2*; wc -l gpurun_out/tunableop_c2*
export PYTORCH_TUNABLEOP_VERBOSE=0
PYTORCH_TUNABLEOP_TUNING=0 $B 2>gpurun_out/tuned_run.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNED', d['value'], d['ms_per_step'])"
PYTORCH_TUNABLEOP_TUNING=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNED', d['value'], d['ms_per_step'])"
