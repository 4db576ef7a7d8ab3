#!/bin/bash
# dev: in-situ A/B of two versions of some kernel sources inside ONE gpurun call: tools/build/csrc_prev/* (older versions of files
# of mmpl_amd/csrc, e.g. from `git show HEAD:...`) against the tree.  usage: tools/src_ab.sh <logname> [bench.py args...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift; mkdir -p $(dirname $out) /tmp/csrc_new; : > $out
for f in tools/build/csrc_prev/*; do cp mmpl_amd/csrc/$(basename $f) /tmp/csrc_new/; done
for v in prev new prev new; do
  if [ $v = prev ]; then cp tools/build/csrc_prev/* mmpl_amd/csrc/; else cp /tmp/csrc_new/* mmpl_amd/csrc/; fi
  python -m mmpl_amd.build > /dev/null 2>&1 || echo "BUILD FAILED $v" >> $out
  timeout 900 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(r['roofline']['achieved'],1), round(r['roofline']['avg_launch_ms'],3))" >> $out
done
cp /tmp/csrc_new/* mmpl_amd/csrc/; python -m mmpl_amd.build > /dev/null 2>&1
cat $out
