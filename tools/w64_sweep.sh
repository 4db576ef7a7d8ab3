#!/bin/bash
# dev: w64 attention kernel -- parity check, same-process A/B against the ping-pong kernel, then timing ablations
# (the kernel is rebuilt on the GPU box with -DW64_ABL=<bits>; see attn_w64.hip)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
timeout 300 python tools/attn_dev.py check 3 4 2>&1 | tail -18 >> $out
timeout 400 python tools/attn_dev.py bench 2 4 2>&1 | grep attn >> $out
for abl in $2; do
  echo "== W64_ABL=$abl" >> $out
  touch mmpl_amd/csrc/attn_w64.hip
  MMPL_EXTRA_HIPCC_FLAGS="-DW64_ABL=$abl" python -m mmpl_amd.build > /dev/null 2>&1
  if [ $((abl & 16)) -ne 0 ]; then
    timeout 200 python tools/attn_dev.py cycles 2>&1 | grep cycles >> $out
  else
    timeout 200 python tools/attn_dev.py bench 4 stages=s1,s3 2>&1 | grep attn >> $out
  fi
done
touch mmpl_amd/csrc/attn_w64.hip; python -m mmpl_amd.build > /dev/null 2>&1
cat $out
