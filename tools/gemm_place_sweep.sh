#!/bin/bash
# dev: gemm_w64 k-loop cycles per placement of the barrier / fragment reads / DMA pieces (each arg: extra -D flags)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out; shift
export MMPL_GEMM_W64=1
for cfg in "$@"; do
  echo "== $cfg" >> $out
  touch mmpl_amd/csrc/gemm_w64.hip
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM_ABL=16 $cfg" python -m mmpl_amd.build 2>&1 | grep -i "error" | head -3 >> $out
  timeout 200 python tools/bench_kernels.py gemmcycles 2>&1 | grep gemmcycles >> $out
done
touch mmpl_amd/csrc/gemm_w64.hip; python -m mmpl_amd.build > /dev/null 2>&1
cat $out
