#!/bin/bash
# dev: GEMM micro-benchmark A/B of two prebuilt libraries (tools/build/libmmpl_hip_{prev,new}.so) in one gpurun call
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out); : > $out
for v in prev new prev new; do
  cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
  echo "== $v" >> $out
  python tools/bench_kernels.py gemm --iters 10 2>&1 | grep "^gemm" | sed 's/|  + split-K.*//' >> $out
done
cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
cat $out
