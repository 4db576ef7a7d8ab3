#!/bin/bash
# dev: A/B of two versions of gemm.hip (tools/build/gemm_prev.hip vs the tree's) in ONE gpurun call: block shapes, then in situ.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $(dirname $out); : > $out
cp mmpl_amd/csrc/gemm.hip /tmp/gemm_new.hip
for v in prev new prev new; do
  if [ $v = prev ]; then cp tools/build/gemm_prev.hip mmpl_amd/csrc/gemm.hip; else cp /tmp/gemm_new.hip mmpl_amd/csrc/gemm.hip; fi
  python -m mmpl_amd.build > /dev/null 2>&1 || echo "BUILD FAILED $v" >> $out
  echo "== $v" >> $out
  timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out
  if [ "$2" = "insitu" ]; then
    timeout 600 python bench.py --steps 8 --warmup 4 --profile-all --no-cpu-baseline --no-vae 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('insitu $v', round(r['sec_per_denoise_step'],4), 'gemm', round(r['gemm_tflops'],1), 'attn', round(r['roofline']['achieved'],1))" >> $out
  fi
done
cp /tmp/gemm_new.hip mmpl_amd/csrc/gemm.hip; python -m mmpl_amd.build > /dev/null 2>&1
cat $out
