#!/bin/bash
# round 4, GPU call 1: suite + default bench line + 16x16x32 mock sweep + heavy-tail line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04a; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q -x > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -5 $out/gputests.log
python -m pytest tests/test_trajectory_gpu.py -m gpu -q -s 2>&1 | grep -E "forwards|steps at|passed|failed" > $out/trajectory.log; cat $out/trajectory.log
python bench.py > $out/bench_14B_720p.json 2> $out/bench.err; tail -c 1500 $out/bench_14B_720p.json
MMPL_CROSS_NO_COLLAPSE=1 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae > $out/bench_14B_720p_no_collapse.json 2>> $out/bench.err
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_14B_720p_profile_all.json 2>> $out/bench.err
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail > $out/bench_14B_720p_heavy_tail.json 2>> $out/bench.err; tail -c 700 $out/bench_14B_720p_heavy_tail.json
bash tools/w64_mock16.sh $tag/mock16.log "3.0:1.5 2.5:1.25 3.0:1.0" 3.0:1.5 > /dev/null 2>&1; cat $out/mock16.log
ls -la $out
