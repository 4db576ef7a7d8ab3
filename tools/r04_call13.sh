#!/bin/bash
# dev: v8 lookahead sweep -- fragment reads every gap instead of every second one, so that the phase-1 barrier (and the DMA of tile t+2) can
# come earlier and the phase-2 wait for tile t+1 later: shader cycles per phase (-DGEMM6_TIMING=1) and wall clock, three block shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out; log=$out/gemm_v8_lookahead_sweep.log; : > $log
for flags in "" "-DGEMM8_RD=1 -DGEMM8_BAR1=18" "-DGEMM8_RD=1 -DGEMM8_BAR2=46" "-DGEMM8_RD=1 -DGEMM8_BAR1=18 -DGEMM8_BAR2=46" "-DGEMM8_RD=1 -DGEMM8_BAR1=18 -DGEMM8_BAR2=40"; do
  echo "== v8 flags [$flags]" >> $log
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_TIMING=1 $flags" python -m mmpl_amd.build > /dev/null 2>&1
  for shape in 25200:15360:5120 25200:5120:13824; do
    BENCH_PHASE_SHAPE=$shape MMPL_GEMM_V8=1 timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep "epi=0" >> $log
  done
  if [ -n "$flags" ]; then MMPL_EXTRA_HIPCC_FLAGS="$flags" python -m mmpl_amd.build > /dev/null 2>&1; else python -m mmpl_amd.build > /dev/null 2>&1; fi
  MMPL_GEMM_V8=1 timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $log
done
python -m mmpl_amd.build > /dev/null 2>&1
echo "== v6" >> $log
MMPL_GEMM_V8=0 timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $log
cat $log
