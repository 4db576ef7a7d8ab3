#!/bin/bash
# dev: same-box sweep of the w64 attention kernel's ring depth and timing ablations (tools/attn_dev.py bench)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
for cfg in "MMPL_W64_RING=4" "MMPL_W64_RING=3" "MMPL_W64_RING=2" "MMPL_W64_ABL=1" "MMPL_W64_ABL=2" "MMPL_W64_ABL=4" "MMPL_W64_ABL=6" "MMPL_W64_ABL=7"; do
  echo "== $cfg" >> $out
  env $cfg timeout 200 python tools/attn_dev.py bench 3 stages=s1,s3 2>&1 | grep attn >> $out
done
MMPL_W64_RING=2 timeout 200 python tools/attn_dev.py check 3 2>&1 | tail -3 >> $out
cat $out
