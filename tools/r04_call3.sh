#!/bin/bash
# round 4, GPU call 3: the three tests that failed in call 2 (i2v image stream fixed; wavefront bit-identity bisected), the fused VAE norm,
# gemm_bf16_v8_kernel: bit-identity with v6 + test_gemm* under MMPL_GEMM_V8=1 + wall clock next to v6 and the vendor yardstick, and the
# vendor kernel under the same counters as v6
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04c; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_fullsize_gpu.py::test_i2v_model_type_production_dims tests/test_i2v_clip_gpu.py tests/test_vae_gpu.py -m gpu -q -s 2>&1 | grep -E "rel_l2|passed|failed|Error" > $out/tests_fixed.log; cat $out/tests_fixed.log
for env in "X=0" "MMPL_ATTN_V1=1" "MMPL_CROSS_NO_COLLAPSE=1" "MMPL_ATTN_NOSPLIT=1" "MMPL_ATTN_NO_MERGE=1"; do
  echo "== $env" >> $out/wavefront_bisect.log
  env $env python -m pytest tests/test_wavefront_gpu.py -m gpu -q -k real_pipeline 2>&1 | grep -E "differ|passed|failed" >> $out/wavefront_bisect.log
done
cat $out/wavefront_bisect.log
MMPL_GEMM_V8=0 timeout 600 python tools/gemm_v8_check.py > $out/gemm_v6_sha.log 2>&1
MMPL_GEMM_V8=1 timeout 600 python tools/gemm_v8_check.py > $out/gemm_v8_sha.log 2>&1
diff $out/gemm_v6_sha.log $out/gemm_v8_sha.log > $out/gemm_v8_vs_v6.diff && echo "v8 == v6 bit for bit" | tee -a $out/gemm_v8_vs_v6.diff; head -30 $out/gemm_v8_vs_v6.diff
MMPL_GEMM_V8=1 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -3 > $out/test_gemm_v8.log; cat $out/test_gemm_v8.log
for v in 0 1 0 1; do
  echo "== MMPL_GEMM_V8=$v" >> $out/gemm_wallclock.log
  MMPL_GEMM_V8=$v timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out/gemm_wallclock.log
done
timeout 300 python tools/bench_kernels.py gemmref --iters 5 2>&1 | grep "^vendor" >> $out/gemm_wallclock.log
cat $out/gemm_wallclock.log
for v in 0 1; do
  echo "== MMPL_VAE_NO_FUSE_NORM=$v" >> $out/vae_fuse_ab.log
  MMPL_VAE_NO_FUSE_NORM=$v python tools/vae_one.py 720p >> $out/vae_fuse_ab.log 2>&1
  MMPL_VAE_NO_FUSE_NORM=$v python tools/vae_one.py 720p >> $out/vae_fuse_ab.log 2>&1
done
cat $out/vae_fuse_ab.log
bash tools/r04_vendor_gemm_pmc.sh $tag/vendor_pmc > /dev/null 2>&1; cat $out/vendor_pmc/summary.txt
for v in 0 1; do
  MMPL_GEMM_V8=$v python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_profile_all_v8_$v.json 2>> $out/bench.err
  python - <<PY
import json; d = json.loads(open("$out/bench_profile_all_v8_$v.json").read().strip().splitlines()[-1])
print("V8=$v step", d["sec_per_denoise_step"], d["sec_per_denoise_step_by_stage"], "gemm", d.get("gemm_tflops"), "attn", d["roofline"]["achieved"])
PY
done
ls -la $out
