#!/bin/bash
# Collects the judged profiles of a build on the GPU box into gpurun_out/prof_<tag>/ (copy the summaries into profiles/):
#   bash tools/r03_profiles.sh <tag> [stats] [hbm] [busy]
# Unlike round 2's script the HBM-traffic passes wrap bench.py ITSELF (the timed command), not a one-op helper: FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 section), summed over every self-attention launch of the run.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
what="${*:-stats hbm busy}"
out=gpurun_out/prof_$tag; mkdir -p $out
for w in $what; do
  case $w in
  stats)
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae > $out/bench_under_rocprof.json 2> $out/stats.err
    ;;
  hbm)
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/hbm_$ctr -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-vae --no-profile --eager > $out/hbm_$ctr.json 2> $out/hbm_$ctr.err
    done
    ;;
  busy)
    timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/busy -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-vae --no-profile > $out/busy_bench.json 2> $out/busy.err
    ;;
  esac
done
python3 tools/r03_profiles_post.py $out $tag
# the raw traces are large: keep the summaries only
find $out -name "*kernel_trace.csv" -size +8M -delete
find $out -name "*counter_collection.csv" -size +8M -delete
ls -la $out
