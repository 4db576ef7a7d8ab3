#!/bin/bash
# dev: gemm_w64 k-loop cycle counts per ablation (timing builds rebuilt on the GPU box with -DGEMM_ABL=<bits>)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
for abl in $2; do
  echo "== GEMM_ABL=$abl" >> $out
  touch mmpl_amd/csrc/gemm_w64.hip
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM_ABL=$abl $GEMM_EXTRA" python -m mmpl_amd.build > /dev/null 2>&1
  if [ $(( ${abl%% *} & 32)) -ne 0 ]; then timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep gemmphases >> $out
  else timeout 200 python tools/bench_kernels.py gemmcycles 2>&1 | grep gemmcycles >> $out; fi
done
touch mmpl_amd/csrc/gemm_w64.hip; python -m mmpl_amd.build > /dev/null 2>&1
cat $out
