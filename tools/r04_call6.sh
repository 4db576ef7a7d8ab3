#!/bin/bash
# round 4, GPU call 6: the sub-tile launch for the partial last round of short-K GEMMs (bit-identity test, per-shape timing on the Wan 1.3B /
# 480p block shapes, in-situ A/B on BASELINE configs[1]), the trajectory tests incl. the committed GPU-semantics fixture
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04f; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py tests/test_trajectory_gpu.py -m gpu -q -s -k "gemm or 50_steps or 2_steps" 2>&1 | grep -E "forwards|steps at|passed|failed|Error|assert" > $out/gemm_traj_tests.log; cat $out/gemm_traj_tests.log
export BENCH_SHAPES="qkv_s1:10920:4608:1536:0,o_s1:10920:1536:1536:3,ffn0_s1:10920:8960:1536:1,ffn2_s1:10920:1536:8960:3,qkv_s0:3120:4608:1536:0,o_s0:3120:1536:1536:3,qkv_s2:9360:4608:1536:0,o_s2:9360:1536:1536:3"
for v in 1 0 1 0; do
  echo "== MMPL_GEMM_NO_SUBTILE=$v" >> $out/gemm_subtile_1p3B_shapes.log
  MMPL_GEMM_NO_SUBTILE=$v timeout 300 python tools/bench_kernels.py gemm --iters 20 2>&1 | grep "^gemm" >> $out/gemm_subtile_1p3B_shapes.log
done
timeout 300 python tools/bench_kernels.py gemmref --iters 20 2>&1 | grep "^vendor" >> $out/gemm_subtile_1p3B_shapes.log
unset BENCH_SHAPES
cat $out/gemm_subtile_1p3B_shapes.log
for v in 1 0 1 0; do
  MMPL_GEMM_NO_SUBTILE=$v python bench.py --model 1.3B --res 480p --steps 16 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_tmp.json 2>> $out/bench.err
  python - <<PY | tee -a $out/bench_1p3B_480p_subtile_ab.log
import json; d = json.loads(open("$out/bench_tmp.json").read().strip().splitlines()[-1])
print("NO_SUBTILE=$v step", round(d["sec_per_denoise_step"], 5), [round(x, 5) for x in d["sec_per_denoise_step_by_stage"]], "gemm", round(d.get("gemm_tflops"), 1), "attn", round(d["roofline"]["achieved"], 1), d.get("kernel_time_share"))
PY
done
ls $out
