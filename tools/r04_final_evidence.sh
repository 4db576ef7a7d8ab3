#!/bin/bash
# One gpurun call that collects the judged evidence of the current build under gpurun_out/<tag>/ and gpurun_out/prof_<tag>/:
#   GPU suite, default bench line, --profile-all line, rocprofv3 stats + PMC passes (tools/r03_profiles.sh), measured full chunks,
#   the other BASELINE configurations.   usage: bash tools/r04_final_evidence.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -3 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python bench.py > $out/bench_14B_720p.json 2> $out/bench.err; tail -c 600 $out/bench_14B_720p.json
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --profile-all > $out/bench_14B_720p_profile_all.json 2>> $out/bench.err
bash tools/r03_profiles.sh $tag stats hbm busy > $out/profiles.log 2>&1; tail -5 $out/profiles.log
python tools/full_chunk.py --model 14B --res 720p > $out/full_chunk_14B_720p.json 2>> $out/bench.err; tail -c 400 $out/full_chunk_14B_720p.json
python bench.py --model 1.3B --res 480p --no-cpu-baseline --profile-all > $out/bench_1p3B_480p.json 2>> $out/bench.err
python tools/full_chunk.py --model 1.3B --res 480p > $out/full_chunk_1p3B_480p.json 2>> $out/bench.err
python bench.py --model 14B --res 480p --no-cpu-baseline > $out/bench_14B_480p.json 2>> $out/bench.err
python bench.py --model 14B --res 720p --mode i2v --i2v-model --no-cpu-baseline > $out/bench_i2v_model_type_14B_720p.json 2>> $out/bench.err
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain 3 > $out/bench_14B_720p_heavy_tail_x3.json 2>> $out/bench.err
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail > $out/bench_14B_720p_heavy_tail_x8.json 2>> $out/bench.err
ls -la $out
