#!/bin/bash
# dev: like env_ab.sh without --profile-all and appending to the log (several configs in one gpurun call).  usage: tools/env_ab2.sh <logname> "<VAR=val>" [bench args...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; cfg="$2"; shift; shift; mkdir -p $(dirname $out); echo "== base vs $cfg: bench.py $*" >> $out
for v in base alt base alt; do
  if [ $v = base ]; then e="A=1"; else e="$cfg"; fi
  env $e timeout 900 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(r['roofline']['achieved'],1))" >> $out
done
cat $out
