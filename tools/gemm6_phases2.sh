#!/bin/bash
# dev: k-loop shader cycles of gemm_bf16_v6_kernel under build flags / env switches.  usage: gemm6_phases2.sh <log> "<flags>|<env>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out); : > $out
for spec in "$@"; do
  flags="${spec%%|*}"; envs="${spec#*|}"
  echo "== flags '$flags' env '$envs'" >> $out
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_TIMING=1 $flags" python -m mmpl_amd.build > /dev/null 2>&1
  env $envs timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep gemmphases >> $out
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $out
