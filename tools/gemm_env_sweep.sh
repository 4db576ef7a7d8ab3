#!/bin/bash
# dev: A/B of run-time switches of the GEMM launcher on the 14B / 720p block shapes inside ONE gpurun call.
# usage: tools/gemm_env_sweep.sh <logname> "<VAR=val ...>" "<VAR=val ...>" ...      ("" = defaults)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out); : > $out
python -m mmpl_amd.build > /dev/null 2>&1
for cfg in "$@"; do
  echo "== env: '$cfg'" >> $out
  env $cfg timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out
done
cat $out
