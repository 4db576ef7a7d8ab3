#!/bin/bash
# Round 5: ONE script for every gpurun call -- named steps, each writing gpurun_out/<tag>/<step>.log (copy what is judged into profiles/).
#   usage: bash tools/r05_gpu.sh <tag> <step> [<step> ...]
# Steps that rebuild the library with experiment defines (MMPL_EXTRA_HIPCC_FLAGS) restore the plain build when they are done.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
ulimit -c 0                                     # a faulting kernel must not fill the box's disk with core files (it did once: r05b)
clean_cores() { rm -f gpucore.* core.* /tmp/gpucore.* /tmp/core.* 2>/dev/null; }
BIG="qkv:25200:15360:5120:0,o:25200:5120:5120:3,ffn0:25200:13824:5120:1,ffn2:25200:5120:13824:3"
S13="qkv_s1:10920:4608:1536:0,o_s1:10920:1536:1536:3,ffn0_s1:10920:8960:1536:1,ffn2_s1:10920:1536:8960:3,qkv_s2:9360:4608:1536:0,o_s2:9360:1536:1536:3,ffn0_s2:9360:8960:1536:1,ffn2_s2:9360:1536:8960:3,qkv_s0:3120:4608:1536:0,o_s0:3120:1536:1536:3"
build() { MMPL_EXTRA_HIPCC_FLAGS="$1" python -m mmpl_amd.build > $out/build.log 2>&1 || { echo "BUILD FAILED ($1)"; tail -5 $out/build.log; }; }
line() {   # one bench line -> "<label> step .. stages .. attn .. gemm .. redo .."
  python3 -c "
import json,sys
r=json.loads(open('$1').read().strip().splitlines()[-1]); rf=r.get('roofline',{})
print('$2', 'step', round(r['sec_per_denoise_step'],4), [round(x,4) for x in r['sec_per_denoise_step_by_stage']], 'attn', round(rf.get('achieved',0),1), 'frac', round(rf.get('frac',0),4),
      'of_sustained', round(rf.get('frac_of_sustained',0),4), 'probe', {k[-12:]:round(v) for k,v in (rf.get('sustained_probe_tflops') or {}).items()}, 'gemm', r.get('gemm_tflops'), 'redo', r.get('attn_blocks_redone_fraction'))"
}
for step in "$@"; do
  echo "=== $step"
  case $step in
    suite)
      python -m pytest tests -m gpu -q -x --deselect tests/test_rccl_loopback_gpu.py > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -4 $out/gputests.log
      python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
    suite_noexit)
      python -m pytest tests -m gpu -q --deselect tests/test_rccl_loopback_gpu.py > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -8 $out/gputests.log
      python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
    rccl)        # last in a call: a transport that hangs must not take the other results with it
      timeout 400 python -m pytest tests/test_rccl_loopback_gpu.py -m gpu -q -s > $out/rccl_loopback.log 2>&1; echo "rc=$?" >> $out/rccl_loopback.log; tail -15 $out/rccl_loopback.log ;;
    newtests)
      python -m pytest tests/test_dit_forward_gpu.py tests/test_bench_multirank_gpu.py tests/test_wavefront_gpu.py tests/test_trajectory_gpu.py -m gpu -q -s > $out/newtests.log 2>&1; echo "rc=$?" >> $out/newtests.log
      grep -E "hand-off for chunk|408 forwards|24 forwards|50 steps|i2v|passed|failed|rc=" $out/newtests.log | tail -40 ;;
    bench)
      python bench.py > $out/bench_14B_720p.json 2> $out/bench.err; line $out/bench_14B_720p.json default ;;
    profall)
      python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_14B_720p_profile_all.json 2>> $out/bench.err; line $out/bench_14B_720p_profile_all.json profile-all ;;
    bench13)
      python bench.py --model 1.3B --res 480p --no-cpu-baseline --profile-all > $out/bench_1p3B_480p.json 2>> $out/bench.err; line $out/bench_1p3B_480p.json 1.3B/480p ;;
    attnref)
      timeout 900 python tools/bench_kernels.py attnref --iters 3 > $out/attn_sdpa_yardstick.log 2>&1; cat $out/attn_sdpa_yardstick.log ;;
    gemm13)      # the 1.3B / 480p block shapes: default launcher, every GEMM on the small tiles, the vendor library
      for e in "A=0" "MMPL_GEMM_V2=1" "MMPL_GEMM_V1=1" "A=0"; do echo "== $e" >> $out/gemm13.log; env $e BENCH_SHAPES=$S13 timeout 300 python tools/bench_kernels.py gemm --iters 20 2>&1 | grep "^gemm" >> $out/gemm13.log; done
      BENCH_SHAPES=$(echo $S13 | sed 's/:[0-9]*,/,/g; s/:[0-9]*$//') timeout 300 python tools/bench_kernels.py gemmref --iters 20 2>&1 | grep "^vendor" >> $out/gemm13.log; cat $out/gemm13.log ;;
    gemmM2)      # mock of cond + uncond batched along M: the same four block GEMMs at M = 25200 and M = 50400
      for m in 25200 50400 25200 50400; do echo "== M=$m" >> $out/gemm_m2.log; BENCH_SHAPES=$(echo $BIG | sed "s/25200/$m/g") timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out/gemm_m2.log; done; cat $out/gemm_m2.log ;;
    polsweep)    # cache policy of the operand streams (LDS-DMA loads): A / W x {nt, sc1}
      for f in "" "-DGEMM_POLICY_A=1" "-DGEMM_POLICY_W=1" "-DGEMM_POLICY_A=1 -DGEMM_POLICY_W=1" "-DGEMM_POLICY_A=2" "-DGEMM_POLICY_W=2" ""; do
        build "$f"; echo "== flags [$f]" >> $out/gemm_policy.log
        BENCH_SHAPES=$BIG timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out/gemm_policy.log
      done; build ""; cat $out/gemm_policy.log ;;
    vb128)       # attention: V fragments by one ds_read_b128 (timing mock of a pre-transposed V tile), in situ, alternating with the shipping build
      for f in "" "-DW64_ABL=256" "" "-DW64_ABL=256"; do
        build "$f"; python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae > $out/vb128_tmp.json 2>> $out/bench.err; line $out/vb128_tmp.json "[$f]" >> $out/attn_vb128_mock.log
      done; build ""; cat $out/attn_vb128_mock.log ;;
    refoff)      # FAST-pass window: reference offset / lower bound rebalanced, heavy tail x8 and x5 and the default weights
      for f in "" "-DW64_REF_OFFSET=96 -DW64_LMIN_EXP=124" "-DW64_REF_OFFSET=80 -DW64_LMIN_EXP=124"; do
        build "$f"
        for g in 8 5; do python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain $g > $out/refoff_tmp.json 2>> $out/bench.err; line $out/refoff_tmp.json "[$f] heavy-tail x$g" >> $out/attn_fast_window.log; done
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --attn-stats > $out/refoff_tmp.json 2>> $out/bench.err; line $out/refoff_tmp.json "[$f] default weights" >> $out/attn_fast_window.log
      done; build ""; cat $out/attn_fast_window.log ;;
    kerneltests)
      python -m pytest tests/test_kernels_gpu.py tests/test_wavefront_gpu.py tests/test_dit_forward_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s > $out/kerneltests.log 2>&1; echo "rc=$?" >> $out/kerneltests.log
      grep -E "hand-off for chunk|deferred at|passed|failed|rc=|Error" $out/kerneltests.log | tail -20 ;;
    conc)        # cond | uncond forwards as two parallel branches of the step graph: 14B / 720p and 1.3B / 480p, alternating with the default
      for a in "" "--concurrent-cfg" "" "--concurrent-cfg"; do
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae $a > $out/conc_tmp.json 2>> $out/bench.err; line $out/conc_tmp.json "14B/720p [$a]" >> $out/bench_concurrent_cfg_ab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae $a > $out/conc_tmp.json 2>> $out/bench.err; line $out/conc_tmp.json "1.3B/480p [$a]" >> $out/bench_concurrent_cfg_ab.log
      done; cat $out/bench_concurrent_cfg_ab.log ;;
    storepol)    # 1.3B / 480p: plain C stores (may stay in the Infinity Cache for the next kernel) vs the shipping nt stores
      for f in "" "-DGEMM6_STORE=0" "" "-DGEMM6_STORE=0"; do
        build "$f"; python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae --profile-all > $out/sp_tmp.json 2>> $out/bench.err; line $out/sp_tmp.json "1.3B/480p [$f]" >> $out/gemm_store_policy_1p3B.log
      done
      build "-DGEMM6_STORE=0"; python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/sp_tmp.json 2>> $out/bench.err; line $out/sp_tmp.json "14B/720p [-DGEMM6_STORE=0]" >> $out/gemm_store_policy_1p3B.log
      build ""; python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/sp_tmp.json 2>> $out/bench.err; line $out/sp_tmp.json "14B/720p []" >> $out/gemm_store_policy_1p3B.log
      cat $out/gemm_store_policy_1p3B.log ;;
    refoff3)     # FAST window (shared reference): offset 64 / 72 / 80 / 88 with the lower bound at 2^-124
      for o in 64 72 80 88; do
        build "-DW64_REF_OFFSET=$o -DW64_LMIN_EXP=124"
        for g in 8 5; do python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain $g > $out/refoff_tmp.json 2>> $out/bench.err; line $out/refoff_tmp.json "offset $o heavy-tail x$g" >> $out/attn_fast_window_offsets.log; done
      done; build ""; cat $out/attn_fast_window_offsets.log ;;
    conc2)       # the shipped rule (concurrent CFG branches per stage where rows x dim <= 48 M) against never / always: 14B at 720p and 480p, 1.3B
      for cfg in "--model 14B --res 720p --steps 8 --warmup 4" "--model 14B --res 480p --steps 8 --warmup 4" "--model 1.3B --res 480p --steps 16 --warmup 8"; do
        for a in "--no-concurrent-cfg" "" "--concurrent-cfg" "--no-concurrent-cfg" ""; do
          python bench.py $cfg --no-cpu-baseline --no-vae $a > $out/conc_tmp.json 2>> $out/bench.err; line $out/conc_tmp.json "[$cfg] [$a]" >> $out/bench_concurrent_cfg_rule.log
        done
      done; cat $out/bench_concurrent_cfg_rule.log ;;
    pipetests)
      python -m pytest tests/test_pipeline_gpu.py tests/test_dit_forward_gpu.py tests/test_trajectory_gpu.py tests/test_i2v_pipeline_gpu.py tests/test_cfg_pair_gpu.py -m gpu -q -s > $out/pipetests.log 2>&1; echo "rc=$?" >> $out/pipetests.log
      grep -E "408 forwards|24 forwards|50 steps|i2v|redo counters|passed|failed|rc=|Error" $out/pipetests.log | tail -30 ;;
    chunk13)
      python tools/full_chunk.py --model 1.3B --res 480p > $out/full_chunk_1p3B_480p.json 2>> $out/bench.err; tail -c 500 $out/full_chunk_1p3B_480p.json ;;
    elem)        # the HBM-bound row passes, two PREBUILT libraries (tools/build/libmmpl_hip_{prev,new}.so) alternating: standalone at the
                 # stage shapes with hashes of the outputs (tools/bench_kernels.py elem), the kernel tests, then in situ with --profile-all
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        timeout 600 python tools/bench_kernels.py elem --iters 20 2>&1 | grep "^elem" > $out/elem_$v.log; sed "s/^/lib=$v /" $out/elem_$v.log >> $out/elem_ab.log
      done
      for v in prev new; do sed -E 's/ +[0-9.]+ us +[0-9.]+ TB.s//' $out/elem_$v.log > $out/elem_sha_$v.txt; done
      diff $out/elem_sha_prev.txt $out/elem_sha_new.txt > $out/elem_sha.diff && echo "outputs of the two libraries: identical hashes on every line" >> $out/elem_ab.log || { echo "HASHES DIFFER:" >> $out/elem_ab.log; cat $out/elem_sha.diff >> $out/elem_ab.log; }
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm or qknorm" 2>&1 | tail -3 >> $out/elem_ab.log
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/el_tmp.json 2>> $out/bench.err; line $out/el_tmp.json "14B/720p [lib=$v]" >> $out/elem_ab.log
        python3 -c "
import json; r=json.loads(open('$out/el_tmp.json').read().strip().splitlines()[-1]); print('   shares', r.get('kernel_time_share'))" >> $out/elem_ab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae > $out/el_tmp.json 2>> $out/bench.err; line $out/el_tmp.json "1.3B/480p [lib=$v]" >> $out/elem_ab.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      cat $out/elem_ab.log ;;
    cross)       # text cross-attention alone (tools/bench_kernels.py cross), two PREBUILT libraries alternating, output hashes; attention tests
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        timeout 600 python tools/bench_kernels.py cross --iters 20 2>&1 | grep "^cross" > $out/cross_$v.log; sed "s/^/lib=$v /" $out/cross_$v.log >> $out/cross_ab.log
      done
      for v in prev new; do sed -E 's/ +[0-9.]+ us .*TFLOP.s//' $out/cross_$v.log > $out/cross_sha_$v.txt; done
      diff $out/cross_sha_prev.txt $out/cross_sha_new.txt > $out/cross_sha.diff && echo "outputs of the two libraries: identical hashes on every line" >> $out/cross_ab.log || { echo "HASHES DIFFER:" >> $out/cross_ab.log; cat $out/cross_sha.diff >> $out/cross_ab.log; }
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_forward_gpu.py -q -m gpu -k "attention or attn or cross or clip" 2>&1 | tail -3 >> $out/cross_ab.log
      cat $out/cross_ab.log ;;
    crossab)     # attn_cross_kernel in situ, two PREBUILT libraries alternating (prev = the commit before): 14B / 720p --profile-all, 1.3B / 480p
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_forward_gpu.py -q -m gpu -k "attention or attn or cross or clip" 2>&1 | tail -3 >> $out/crossab.log
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/cx_tmp.json 2>> $out/bench.err; line $out/cx_tmp.json "14B/720p [lib=$v]" >> $out/crossab.log
        python3 -c "
import json; r=json.loads(open('$out/cx_tmp.json').read().strip().splitlines()[-1]); print('   shares', r.get('kernel_time_share'))" >> $out/crossab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae > $out/cx_tmp.json 2>> $out/bench.err; line $out/cx_tmp.json "1.3B/480p [lib=$v]" >> $out/crossab.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      cat $out/crossab.log ;;
    insitu)      # two PREBUILT libraries alternating in situ: 14B / 720p --profile-all (with shares) and 1.3B / 480p
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/is_tmp.json 2>> $out/bench.err; line $out/is_tmp.json "14B/720p [lib=$v]" >> $out/insitu_ab.log
        python3 -c "
import json; r=json.loads(open('$out/is_tmp.json').read().strip().splitlines()[-1]); print('   shares', r.get('kernel_time_share'))" >> $out/insitu_ab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae > $out/is_tmp.json 2>> $out/bench.err; line $out/is_tmp.json "1.3B/480p [lib=$v]" >> $out/insitu_ab.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      cat $out/insitu_ab.log ;;
    libab)       # two PREBUILT libraries (tools/build/libmmpl_hip_{prev,new}.so) in situ, alternating: 14B / 720p and 1.3B / 480p; then bit-identity of the GEMMs
      for v in prev new prev new; do
        cp tools/build/libmmpl_hip_$v.so mmpl_amd/lib/libmmpl_hip.so
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/ab_tmp.json 2>> $out/bench.err; line $out/ab_tmp.json "14B/720p lib=$v" >> $out/lib_ab.log
        python bench.py --model 1.3B --res 480p --steps 16 --warmup 8 --no-cpu-baseline --no-vae > $out/ab_tmp.json 2>> $out/bench.err; line $out/ab_tmp.json "1.3B/480p lib=$v" >> $out/lib_ab.log
        timeout 600 python tools/gemm_v8_check.py > $out/gemm_sha_$v.log 2>&1
        BENCH_SHAPES=$BIG timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed "s/^/lib=$v /" >> $out/lib_ab_gemm_standalone.log
      done
      cp tools/build/libmmpl_hip_new.so mmpl_amd/lib/libmmpl_hip.so
      diff $out/gemm_sha_prev.log $out/gemm_sha_new.log > $out/gemm_sha.diff && echo "GEMM outputs of the two libraries: identical hashes on every line" | tee -a $out/lib_ab.log
      cat $out/lib_ab.log $out/lib_ab_gemm_standalone.log ;;
    profiles)    # rocprofv3 kernel stats + PMC passes over bench.py itself (tools/r03_profiles.sh), summaries -> gpurun_out/prof_<tag>/
      bash tools/r03_profiles.sh $tag stats hbm busy > $out/profiles.log 2>&1; tail -6 $out/profiles.log ;;
    profstats)   # rocprofv3 --kernel-trace --stats over bench.py itself (tools/r03_profiles.sh), summary -> gpurun_out/prof_<tag>/
      bash tools/r03_profiles.sh $tag stats > $out/profiles.log 2>&1; tail -6 $out/profiles.log ;;
    profhbm)     # FETCH_SIZE / WRITE_SIZE PMC passes over bench.py itself (tools/r03_profiles.sh): attention op + per-kernel table
      bash tools/r03_profiles.sh $tag hbm > $out/profiles_hbm.log 2>&1; tail -6 $out/profiles_hbm.log ;;
    chunk14)
      python tools/full_chunk.py --model 14B --res 720p > $out/full_chunk_14B_720p.json 2>> $out/bench.err; tail -c 500 $out/full_chunk_14B_720p.json ;;
    others)      # the other BASELINE / reported configurations
      python bench.py --model 14B --res 480p --no-cpu-baseline > $out/bench_14B_480p.json 2>> $out/bench.err; line $out/bench_14B_480p.json "14B/480p"
      python bench.py --model 14B --res 720p --mode i2v --i2v-model --no-cpu-baseline > $out/bench_i2v_model_type_14B_720p.json 2>> $out/bench.err; line $out/bench_i2v_model_type_14B_720p.json "I2V model type 14B/720p"
      python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain 3 > $out/bench_14B_720p_heavy_tail_x3.json 2>> $out/bench.err; line $out/bench_14B_720p_heavy_tail_x3.json "heavy tail x3"
      python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail > $out/bench_14B_720p_heavy_tail_x8.json 2>> $out/bench.err; line $out/bench_14B_720p_heavy_tail_x8.json "heavy tail x8"
      python3 -c "
import json
for f in ('x3','x8'):
    r=json.loads(open('$out/bench_14B_720p_heavy_tail_'+f+'.json').read().strip().splitlines()[-1]); print(f, 'blocks redone', r.get('attn_blocks_redone_fraction'), 'waves failed', r.get('attn_waves_failed_fraction'))" ;;
    n2gloo)      # the DEFAULT N = 2 line (a measured wavefront) on one GPU over gloo: functional, Wan 1.3B at 480p
      timeout 1500 python bench.py --gpus 2 --dist-backend gloo --model 1.3B --res 480p --wavefront-budget-s 240 > $out/bench_default_n2_gloo_1p3B_480p.json 2>> $out/bench.err; tail -c 1500 $out/bench_default_n2_gloo_1p3B_480p.json ;;
    shareab)     # block 0's self-attention computed once per step (default) vs by both branches: 14B / 720p and 14B / 480p, alternating
      for a in "--no-share-block0" "" "--no-share-block0" ""; do
        python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae $a > $out/sh_tmp.json 2>> $out/bench.err; line $out/sh_tmp.json "14B/720p [$a]" >> $out/bench_share_block0_ab.log
        python bench.py --res 480p --steps 8 --warmup 4 --no-cpu-baseline --no-vae $a > $out/sh_tmp.json 2>> $out/bench.err; line $out/sh_tmp.json "14B/480p [$a]" >> $out/bench_share_block0_ab.log
      done; cat $out/bench_share_block0_ab.log ;;
    groupsweep)  # M-tile group of the block order (MMPL_GEMM_GROUP) on the four 14B / 720p block shapes, at M = 25200 and 21600
      for m in 25200 21600; do for g in 0 2 3 4 6 8; do
        echo "== M=$m MMPL_GEMM_GROUP=$g (0 = the launcher's choice)" >> $out/gemm_group_sweep.log
        MMPL_GEMM_GROUP=$g BENCH_SHAPES=$(echo $BIG | sed "s/25200/$m/g") timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $out/gemm_group_sweep.log
      done; done; cat $out/gemm_group_sweep.log ;;
    n2gloo14)    # the DEFAULT N = 2 line at 14B scale (480p so that two ranks fit one GPU), gloo: the budgeted step count and the scaling at a realistic size
      timeout 1500 python bench.py --gpus 2 --dist-backend gloo --model 14B --res 480p --no-cpu-baseline > $out/bench_default_n2_gloo_14B_480p.json 2>> $out/bench.err; tail -c 1800 $out/bench_default_n2_gloo_14B_480p.json ;;
    v8all)       # v8 (now with the second half's residual rows in flight under the first half's stores) on EVERY large GEMM vs the launcher's choice (v8 for N >= 8192 only)
      for e in "A=0" "MMPL_GEMM_V8=1" "A=0" "MMPL_GEMM_V8=1"; do
        env $e python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/v8_tmp.json 2>> $out/bench.err; line $out/v8_tmp.json "14B/720p [$e]" >> $out/gemm_v8_everywhere_ab.log
        echo "== $e standalone" >> $out/gemm_v8_everywhere_ab.log; env $e BENCH_SHAPES=$BIG timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $out/gemm_v8_everywhere_ab.log
      done; cat $out/gemm_v8_everywhere_ab.log ;;
    *) echo "unknown step $step" ;;
  esac
  clean_cores
done
ls -la $out
