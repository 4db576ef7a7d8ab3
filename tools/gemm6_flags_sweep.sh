#!/bin/bash
# dev: A/B of compile-time variants of gemm.hip on the 14B / 720p block shapes, all inside ONE gpurun call (boxes differ by ~5 %).
# usage: tools/gemm6_flags_sweep.sh <logname> "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the shipping build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out); : > $out
for flags in "$@"; do
  echo "== flags: '$flags'" >> $out
  MMPL_EXTRA_HIPCC_FLAGS="$flags" python -m mmpl_amd.build > /dev/null 2>&1 || echo "BUILD FAILED" >> $out
  timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $out
