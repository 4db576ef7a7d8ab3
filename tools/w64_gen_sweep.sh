#!/bin/bash
# dev: cycle count of the steady loop per generator setting (each argument: "VAR=val VAR=val ..." for tools/gen_attn_w64.py),
# timing build rebuilt on the GPU box; the first argument names the log.  A setting prefixed with "check:" also runs the parity check.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out; shift
for cfg in "$@"; do
  chk=0; case "$cfg" in check:*) chk=1; cfg="${cfg#check:}";; esac
  echo "== $cfg" >> $out
  env $cfg python tools/gen_attn_w64.py 2>&1 | tail -1 >> $out
  if [ $chk = 1 ]; then
    python -m mmpl_amd.build > /dev/null 2>&1
    timeout 300 python tools/attn_dev.py check 4 2>&1 | tail -3 >> $out
  fi
  # (build.py keys staleness on the flags too: switching to / from the timing build always recompiles)
  MMPL_EXTRA_HIPCC_FLAGS="-DW64_ABL=16" python -m mmpl_amd.build > /dev/null 2>&1
  timeout 200 python tools/attn_dev.py cycles 2>&1 | grep cycles >> $out
done
# restore: the committed default schedule and a clean library (the timing build's attention outputs are cycle counts, not O)
python tools/gen_attn_w64.py > /dev/null 2>&1
python -m mmpl_amd.build > /dev/null 2>&1
cat $out
