cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ref_trace
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/ref_trace -o ref -- python bench.py --workload REF --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/ref_trace.log 2>&1
tail -2 gpurun_out/ref_trace.log | head -c 600
find gpurun_out/ref_trace -name "*.db" | head
python tools/dbg/step_kernels.py $(find gpurun_out/ref_trace -name "*.db" | head -1) 45 --gaps
