#!/bin/bash
# dev: v8 with the loop-top fragment wait split per activation fragment (GEMM8_FINEWAIT) vs one lgkmcnt(0): bit-identity, cycles, wall clock
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out; log=$out/gemm_v8_finewait.log; : > $log
MMPL_GEMM_V8=0 timeout 600 python tools/gemm_v8_check.py > $out/fw_v6_sha.log 2>&1
MMPL_GEMM_V8=1 timeout 600 python tools/gemm_v8_check.py > $out/fw_v8_sha.log 2>&1
diff $out/fw_v6_sha.log $out/fw_v8_sha.log > /dev/null && echo "v8 (fine waits) == v6 bit for bit" >> $log || echo "MISMATCH v8 vs v6" >> $log
for fw in 1 0 1 0; do
  echo "== GEMM8_FINEWAIT=$fw" >> $log
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_TIMING=1 -DGEMM8_FINEWAIT=$fw" python -m mmpl_amd.build > /dev/null 2>&1
  BENCH_PHASE_SHAPE=25200:15360:5120 MMPL_GEMM_V8=1 timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep "epi=0" >> $log
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM8_FINEWAIT=$fw" python -m mmpl_amd.build > /dev/null 2>&1
  MMPL_GEMM_V8=1 timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $log
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $log
