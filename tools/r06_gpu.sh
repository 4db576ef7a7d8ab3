#!/bin/bash
# ONE script for every gpurun call of the round -- named steps, each writing gpurun_out/<tag>/<step>.log (copy what is judged into
# profiles/).    usage: bash tools/r06_gpu.sh <tag> <step>[:<arg>] [<step> ...]
# Development builds (experiment defines) go through `build "-DMMPL_DEV_ABLATIONS -D..."` and restore the plain build when done.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
ulimit -c 0                                     # a faulting kernel must not fill the box's disk with core files
BIG="qkv:25200:15360:5120:0,o:25200:5120:5120:3,ffn0:25200:13824:5120:1,ffn2:25200:5120:13824:3"
SHORT="--steps 8 --warmup 4 --no-cpu-baseline --no-vae"
build() { MMPL_EXTRA_HIPCC_FLAGS="$1" python -m mmpl_amd.build > $out/build.log 2>&1 || { echo "BUILD FAILED ($1)"; tail -5 $out/build.log; }; }
line() {   # one bench line -> "<label> step .. stages .. attn .. gemm .. redo / predicted .."
  python3 -c "
import json,sys
r=json.loads(open('$1').read().strip().splitlines()[-1]); rf=r.get('roofline',{})
print('$2', 'step', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(rf.get('achieved',0),1), 'frac', round(rf.get('frac',0),4),
      'of_sustained', round(rf.get('frac_of_sustained',0),4), 'probe', {k[-12:]:round(v) for k,v in (rf.get('sustained_probe_tflops') or {}).items()}, 'gemm', r.get('gemm_tflops'),
      'redo', r.get('attn_blocks_redone_fraction'), 'predicted', r.get('attn_blocks_predicted_fraction'), 'remembered', r.get('attn_blocks_fast_on_remembered_reference_fraction'), 'shares', r.get('kernel_time_share'))"
}
for spec in "$@"; do
  step=${spec%%:*}; arg=${spec#*:}; [ "$arg" = "$spec" ] && arg=""
  echo "=== $step $arg"
  case $step in
    suite)       # what the driver runs at round end
      python -m pytest tests -m gpu -q -x --deselect tests/test_rccl_loopback_gpu.py > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -4 $out/gputests.log
      python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
    suite_noexit)
      python -m pytest tests -m gpu -q --deselect tests/test_rccl_loopback_gpu.py > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -12 $out/gputests.log
      python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
    rccl)        # last in a call: a transport that hangs must not take the other results with it
      timeout 400 python -m pytest tests/test_rccl_loopback_gpu.py -m gpu -q -s > $out/rccl_loopback.log 2>&1; echo "rc=$?" >> $out/rccl_loopback.log; tail -15 $out/rccl_loopback.log ;;
    tests)       # tests:<pytest selection>, e.g. tests:"tests/test_attn_history_gpu.py tests/test_kernels_gpu.py -k attention"
      n=$(ls $out/tests*.log 2>/dev/null | wc -l)
      timeout 2400 python -m pytest $arg -m gpu -q -s > $out/tests$n.log 2>&1; echo "rc=$? ($arg)" >> $out/tests$n.log
      grep -E "passed|failed|rc=|Error|error|rel|redo|408 forwards|chunk" $out/tests$n.log | tail -40 ;;
    bench)
      python bench.py > $out/bench_14B_720p.json 2> $out/bench.err; line $out/bench_14B_720p.json default ;;
    profall)
      python bench.py $SHORT --profile-all > $out/bench_14B_720p_profile_all.json 2>> $out/bench.err; line $out/bench_14B_720p_profile_all.json profile-all ;;
    bench13)
      python bench.py --model 1.3B --res 480p --no-cpu-baseline --profile-all > $out/bench_1p3B_480p.json 2>> $out/bench.err; line $out/bench_1p3B_480p.json 1.3B/480p ;;
    bench480)
      python bench.py --res 480p $SHORT > $out/bench_14B_480p.json 2>> $out/bench.err; line $out/bench_14B_480p.json 14B/480p ;;
    chunk)
      python tools/full_chunk.py > $out/full_chunk_14B_720p.json 2>> $out/bench.err; tail -c 600 $out/full_chunk_14B_720p.json ;;
    chunk13)
      python tools/full_chunk.py --model 1.3B --res 480p > $out/full_chunk_1p3B_480p.json 2>> $out/bench.err; tail -c 600 $out/full_chunk_1p3B_480p.json ;;
    hist)        # the pass history of the self-attention (item 1): default weights, heavy tail x8 on every head, x8 on 15 % of the heads;
                 # history on / off alternating on ONE box
      for w in "" "--heavy-tail" "--heavy-tail --heavy-tail-heads 0.15"; do
        for h in "" "--no-attn-history" "" "--no-attn-history"; do
          python bench.py $SHORT --attn-stats $w $h > $out/hist_tmp.json 2>> $out/bench.err; line $out/hist_tmp.json "[$w] [$h]" >> $out/attn_history_ab.log
        done
      done; cat $out/attn_history_ab.log ;;
    ht)          # the heavy-tail lines kept under profiles/: x3, x8, x8 on 15 % of the heads (20 steps like the default line)
      python bench.py --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain 3 > $out/bench_14B_720p_heavy_tail_x3.json 2>> $out/bench.err; line $out/bench_14B_720p_heavy_tail_x3.json x3
      python bench.py --no-cpu-baseline --no-vae --heavy-tail > $out/bench_14B_720p_heavy_tail_x8.json 2>> $out/bench.err; line $out/bench_14B_720p_heavy_tail_x8.json x8
      python bench.py --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-heads 0.15 > $out/bench_14B_720p_heavy_tail_x8_p15.json 2>> $out/bench.err; line $out/bench_14B_720p_heavy_tail_x8_p15.json "x8 p=0.15" ;;
    htlong)      # heavy tail x8 over 40 timed steps (every stage replayed 12 times): the history's steady state, on / off
      for h in "" "--no-attn-history"; do
        python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-vae --heavy-tail $h > $out/htlong_tmp.json 2>> $out/bench.err; line $out/htlong_tmp.json "x8, 40 steps [$h]" >> $out/heavy_tail_x8_40_steps.log
      done; cat $out/heavy_tail_x8_40_steps.log ;;
    i2vmodel)
      python bench.py $SHORT --i2v-model > $out/bench_i2v_model_type_14B_720p.json 2>> $out/bench.err; line $out/bench_i2v_model_type_14B_720p.json "Wan-I2V model type 14B/720p" ;;
    insitu)      # two PREBUILT libraries alternating in situ (tools/build/libmmpl_hip_{prev,new}.so): 14B / 720p --profile-all and 1.3B / 480p
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        python bench.py $SHORT --profile-all $arg > $out/is_tmp.json 2>> $out/bench.err; line $out/is_tmp.json "14B/720p [lib=$v]" >> $out/insitu_ab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae $arg > $out/is_tmp.json 2>> $out/bench.err; line $out/is_tmp.json "1.3B/480p [lib=$v]" >> $out/insitu_ab.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so; cat $out/insitu_ab.log ;;
    attnhash)    # do two PREBUILT libraries compute the same self-attention bits (stateless launches)?
      for v in prev new; do cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so; timeout 300 python tools/attn_hash.py 2>&1 | grep "^attnhash" > $out/attnhash_$v.txt; done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      diff $out/attnhash_prev.txt $out/attnhash_new.txt > $out/attnhash.diff && echo "identical hashes on every shape" || cat $out/attnhash.diff; cat $out/attnhash_new.txt ;;
    gemmab)      # the four 14B / 720p block GEMMs standalone, two PREBUILT libraries alternating, output hashes
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        echo "== lib=$v" >> $out/gemm_ab.log; BENCH_SHAPES=$BIG timeout 300 python tools/bench_kernels.py gemm --iters 10 2>&1 | grep "^gemm" >> $out/gemm_ab.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so; cat $out/gemm_ab.log ;;
    devflags)    # devflags:"<flags>": a DEVELOPMENT build of the library (dev_knobs.h) against the plain one on the block GEMMs, alternating
      for f in "" "$arg" "" "$arg"; do
        build "${f:+-DMMPL_DEV_ABLATIONS $f}"; echo "== flags [$f]" >> $out/devflags.log
        BENCH_SHAPES=$BIG timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out/devflags.log
      done; build ""; cat $out/devflags.log ;;
    gemmphases)  # { prologue, k loop, epilogue } shader cycles of the large GEMM kernels (a -DGEMM6_TIMING=1 development build)
      build "-DMMPL_DEV_ABLATIONS -DGEMM6_TIMING=1 $arg"; timeout 300 python tools/bench_kernels.py gemmphases 2>&1 | grep gemmphases > $out/gemm_phases.log; build ""; cat $out/gemm_phases.log ;;
    tcc)         # L2 hit / miss and fabric read bytes of the six block GEMM shapes, one shape per pass (item 4: how much of the fabric traffic is geometry?)
      for shp in qkv:25200:15360:5120:0 o:25200:5120:5120:3 cq:25200:5120:5120:0 co:25200:5120:5120:4 ffn0:25200:13824:5120:1 ffn2:25200:5120:13824:3; do
        name=${shp%%:*}
        for pmc in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
          t=$(echo $pmc | tr ' ' '_')
          BENCH_SHAPES=$shp timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $out/tcc_${name}__$t -o p -- python3 tools/bench_kernels.py gemm --iters 1 > /dev/null 2> $out/tcc_${name}__$t.err
        done
      done
      python3 tools/profiles_post.py $out $tag tcc; cat $out/${tag}_pmc_gemm_l2.md ;;
    stats)       # rocprofv3 kernel stats of the bench command itself
      timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python3 bench.py $SHORT > $out/bench_under_rocprof.json 2> $out/stats.err
      python3 tools/profiles_post.py $out $tag; head -12 $out/${tag}_kernel_stats.csv ;;
    hbm)         # FETCH_SIZE / WRITE_SIZE in separate passes over bench.py itself (MI355X_MICROARCH.md, HBM / rocprofv3 section)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/hbm_$ctr -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-vae --no-profile --eager > $out/hbm_$ctr.json 2> $out/hbm_$ctr.err
      done
      python3 tools/profiles_post.py $out $tag; cat $out/${tag}_pmc_attention_hbm.json ;;
    busy)
      timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/busy -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-vae --no-profile > $out/busy_bench.json 2> $out/busy.err
      python3 tools/profiles_post.py $out $tag; cat $out/${tag}_pmc_mfma_busy.md ;;
    n2)          # the default N = 2 line (measured wavefront; gloo: both ranks on this one GPU -- functional only)
      timeout 1500 python bench.py --gpus 2 --dist-backend gloo --model 1.3B --res 480p > $out/bench_default_n2_gloo_1p3B_480p.json 2>> $out/bench.err; tail -c 1500 $out/bench_default_n2_gloo_1p3B_480p.json ;;
    *) echo "unknown step $step" ;;
  esac
  rm -f gpucore.* core.* /tmp/gpucore.* /tmp/core.* 2>/dev/null
done
# the raw traces are large: keep the summaries only
find $out -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
find $out -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
ls $out | head -60
