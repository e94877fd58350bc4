#!/bin/bash
# Address-sanitizer run of the pointops kernels (run ON THE GPU BOX through gpurun, from the repo root).
# Builds lib_asan/libpcm_pointops.so (`make -C pointcloudmatters_amd/csrc asan`), then runs the small-shape bit-exact pointops
# tests against it with xnack on and the sanitizer runtime preloaded.  Output: gpurun_out/asan_<tag>.log
TAG=${1:-r03}
mkdir -p gpurun_out
LOG=gpurun_out/asan_$TAG.log
{
  echo "# make asan"; make -C pointcloudmatters_amd/csrc -j8 asan 2>&1 | tail -3
  RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
  echo "# runtime: $RT"
  export HSA_XNACK=1 PCM_POINTOPS_LIB=$PWD/pointcloudmatters_amd/lib_asan/libpcm_pointops.so
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1
  echo "# pytest (LD_PRELOAD=$RT)"
  LD_PRELOAD=$RT timeout 900 python -m pytest tests/test_pointops_gpu.py tests/test_segsum_gpu.py -q -m gpu -x -k "not 4096 and not 16384 and not 9000" > gpurun_out/asan_pytest_$TAG.txt 2>&1
  echo "# pytest exit status $?"
  grep -v "NCCL\|RCCL" gpurun_out/asan_pytest_$TAG.txt | tail -25
} > $LOG 2>&1
tail -20 $LOG
