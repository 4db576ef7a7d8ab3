#!/bin/bash
# dev (round 4): the full 16x16x32 timing mock of attn_w64_kernel against the shipping kernel, one gpurun call (same box):
#   shader cycles per KV tile of the steady loop (-DW64_ABL=16 builds) and standalone wall clock on the 14B / 720p stage shapes
#   (tools/attn_dev.py bench 4), per generator budget; then the best budget in situ (bench.py attention ms).  Mock results are garbage.
# usage: bash tools/w64_mock16.sh <logname> "<budget>:<wexp> [<budget>:<wexp> ...]" [insitu <budget>:<wexp>]   (generator settings)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
insitu() {   # $1 = label
  timeout 900 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn TFLOP/s', round(r['roofline']['achieved'],1), 'ms/op', round(r['roofline']['avg_launch_ms'],3))" >> $out
}
echo "== shipping kernel (32x32x16)" >> $out
python -m mmpl_amd.build > /dev/null 2>&1
timeout 300 python tools/attn_dev.py bench 4 stages=s1,s3 2>&1 | grep "^attn" >> $out
MMPL_EXTRA_HIPCC_FLAGS="-DW64_ABL=16" python -m mmpl_amd.build > /dev/null 2>&1
timeout 200 python tools/attn_dev.py cycles 2>&1 | grep cycles >> $out
for b in $2; do
  echo "== mock16 budget $b" >> $out
  W64_BUDGET16=${b%%:*} W64_WEXP=${b##*:} python tools/gen_attn_w64_mock16.py 2>&1 | tail -1 >> $out
  MMPL_EXTRA_HIPCC_FLAGS="-DW64_MOCK16=1 -DW64_ABL=16" python -m mmpl_amd.build > /dev/null 2>&1 || echo "BUILD FAILED" >> $out
  timeout 200 python tools/attn_dev.py cycles 2>&1 | grep cycles >> $out
  MMPL_EXTRA_HIPCC_FLAGS="-DW64_MOCK16=1" python -m mmpl_amd.build > /dev/null 2>&1 || echo "BUILD FAILED" >> $out
  timeout 300 python tools/attn_dev.py bench 4 stages=s1,s3 2>&1 | grep "^attn" >> $out
done
if [ -n "$3" ]; then
  echo "== in situ (bench.py, same box): shipping / mock16 budget $3 / shipping / mock16" >> $out
  for v in ship mock ship mock; do
    if [ $v = ship ]; then python -m mmpl_amd.build > /dev/null 2>&1
    else W64_BUDGET16=${3%%:*} W64_WEXP=${3##*:} python tools/gen_attn_w64_mock16.py > /dev/null 2>&1; MMPL_EXTRA_HIPCC_FLAGS="-DW64_MOCK16=1" python -m mmpl_amd.build > /dev/null 2>&1; fi
    insitu $v
  done
fi
python -m mmpl_amd.build > /dev/null 2>&1
cat $out
