#!/bin/bash
# round 4, GPU call 5: the FAST-pass reference from 4 KV tiles vs 1 (default and heavy-tail x8 lines, same box), v8 (no prefetch) for the wide
# GEMMs vs v6 everywhere in situ, the new trajectory test, full suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04e; out=gpurun_out/$tag; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "step", round(d["sec_per_denoise_step"], 4), [round(x, 4) for x in d["sec_per_denoise_step_by_stage"]], "gemm", d.get("gemm_tflops") and round(d["gemm_tflops"], 1),
      "attn", round(d["roofline"]["achieved"], 1), "redo", d.get("attn_blocks_redone_fraction"))
PY
}
python -m pytest tests/test_trajectory_gpu.py tests/test_kernels_gpu.py -m gpu -q -s -k "trajectory or 50_steps or 2_steps or attention" 2>&1 | grep -E "forwards|steps at|passed|failed|Error" > $out/traj_attn_tests.log; cat $out/traj_attn_tests.log
for v in 0 auto 0 auto 0 auto; do
  if [ $v = auto ]; then unset MMPL_GEMM_V8; else export MMPL_GEMM_V8=$v; fi
  python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_tmp.json 2>> $out/bench.err
  line $out/bench_tmp.json "V8=$v" | tee -a $out/bench_v8_wide_only_insitu_ab.log
done
unset MMPL_GEMM_V8
for rt in 4 1 4 1; do
  if [ $rt = 4 ]; then python -m mmpl_amd.build > /dev/null 2>&1; else MMPL_EXTRA_HIPCC_FLAGS="-DW64_REF_TILES=$rt" python -m mmpl_amd.build > /dev/null 2>&1; fi
  python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae > $out/bench_tmp.json 2>> $out/bench.err
  line $out/bench_tmp.json "W64_REF_TILES=$rt default weights" | tee -a $out/attn_ref_tiles_ab.log
  python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail > $out/bench_tmp.json 2>> $out/bench.err
  line $out/bench_tmp.json "W64_REF_TILES=$rt heavy-tail x8" | tee -a $out/attn_ref_tiles_ab.log
  if [ $rt = 4 ]; then cp $out/bench_tmp.json $out/bench_14B_720p_heavy_tail_x8_ref4.json; fi
done
python -m mmpl_amd.build > /dev/null 2>&1
python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain 5 > $out/bench_14B_720p_heavy_tail_x5_ref4.json 2>> $out/bench.err
line $out/bench_14B_720p_heavy_tail_x5_ref4.json "W64_REF_TILES=4 heavy-tail x5" | tee -a $out/attn_ref_tiles_ab.log
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -6 $out/gputests.log
ls $out
