#!/bin/bash
# dev: in-situ A/B of two PREBUILT libraries inside ONE gpurun call (no compile time on the GPU box): tools/build/libmmpl_hip_prev.so
# (e.g. built from `git stash`) against tools/build/libmmpl_hip_new.so.  usage: tools/lib_ab.sh <logname> [bench.py args...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out); echo "== bench.py $*" >> $out
for v in prev new prev new; do
  cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
  timeout 900 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(r['roofline']['achieved'],1), round(r['roofline']['avg_launch_ms'],3))" >> $out
done
cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
cat $out
