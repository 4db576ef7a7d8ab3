#!/bin/bash
# dev: { prologue, k loop, epilogue } shader cycles per tile of gemm_bf16_v6_kernel and gemm_bf16_v8_kernel (-DGEMM6_TIMING=1) on three block shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out; log=$out/gemm_phases_v6_v8.log; : > $log
MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_TIMING=1" python -m mmpl_amd.build > /dev/null 2>&1
for shape in 25200:5120:5120 25200:15360:5120 25200:5120:13824; do
  for v in 0 1; do
    echo "== shape $shape MMPL_GEMM_V8=$v" >> $log
    BENCH_PHASE_SHAPE=$shape MMPL_GEMM_V8=$v timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep gemmphases >> $log
  done
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $log
