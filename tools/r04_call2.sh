#!/bin/bash
# round 4, GPU call 2: full suite, oracle-on-this-host trajectory, bench lines after the FAST-pass row reference, attention energy ablations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04b; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -5 $out/gputests.log
python -m pytest tests/test_trajectory_gpu.py tests/test_dit_forward_gpu.py -m gpu -q -s 2>&1 | grep -E "forwards|steps at|n_valid|passed|failed" > $out/trajectory.log; cat $out/trajectory.log
(MMPL_FULL_TRAJ=1 python -m pytest tests/test_oracle_golden.py -q -s -k 50_steps 2>&1 | grep -E "oracle vs|passed|failed" > $out/oracle_other_cpu_trajectory.log; cat $out/oracle_other_cpu_trajectory.log) &
python bench.py --no-cpu-baseline > $out/bench_14B_720p.json 2> $out/bench.err; tail -c 900 $out/bench_14B_720p.json
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail > $out/bench_14B_720p_heavy_tail_x8.json 2>> $out/bench.err; tail -c 500 $out/bench_14B_720p_heavy_tail_x8.json
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain 3 > $out/bench_14B_720p_heavy_tail_x3.json 2>> $out/bench.err; tail -c 500 $out/bench_14B_720p_heavy_tail_x3.json
wait
# attention: wall-clock ablations standalone (power-limited chip: time ~ energy per op).  garbage results, timing only
log=$out/attn_wallclock_ablations.log; : > $log
for abl in 0 2 4 1 8 6 7 15 32 64 128; do
  if [ $abl = 0 ]; then python -m mmpl_amd.build > /dev/null 2>&1; else MMPL_EXTRA_HIPCC_FLAGS="-DW64_ABL=$abl" python -m mmpl_amd.build > /dev/null 2>&1; fi
  echo "== W64_ABL=$abl" >> $log
  timeout 300 python tools/attn_dev.py bench 4 stages=s3 2>&1 | grep "^attn" >> $log
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $log
ls -la $out
