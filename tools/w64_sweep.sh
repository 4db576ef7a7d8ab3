#!/bin/bash
# dev: w64 attention kernel -- parity check, then same-process A/B against the ping-pong kernel on the 14B/720p stage shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
timeout 300 python tools/attn_dev.py check 3 4 2>&1 | tail -18 >> $out
timeout 400 python tools/attn_dev.py bench 2 4 2>&1 | grep attn >> $out
cat $out
