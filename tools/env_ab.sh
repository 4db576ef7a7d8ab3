#!/bin/bash
# dev: in-situ A/B of run-time switches (mmpl_config.h) inside ONE gpurun call.  usage: tools/env_ab.sh <logname> "<VAR=val ...>" [bench args...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; cfg="$2"; shift; shift; mkdir -p $(dirname $out); : > $out
python -m mmpl_amd.build > /dev/null 2>&1
for v in base alt base alt; do
  if [ $v = base ]; then e="A=1"; else e="$cfg"; fi
  env $e timeout 900 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(r['roofline']['achieved'],1), 'gemm', round(r['gemm_tflops'],1), r['kernel_time_share'])" >> $out
done
cat $out
