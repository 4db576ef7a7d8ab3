cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04p
export BENCH_SHAPES="ffn0_s1:10920:8960:1536:1,ffn0_s0:3120:8960:1536:1,ffn0_s2:9360:8960:1536:1,ffn0_14B_480p_s1:10920:13824:5120:1,qkv_14B_480p_s1:10920:15360:5120:0"
for v in 0 auto 0 auto; do
  if [ $v = auto ]; then unset MMPL_GEMM_V8; else export MMPL_GEMM_V8=$v; fi
  echo "== MMPL_GEMM_V8=$v" >> gpurun_out/r04p/gemm_v8_auto_other_configs.log
  timeout 300 python tools/bench_kernels.py gemm --iters 20 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> gpurun_out/r04p/gemm_v8_auto_other_configs.log
done
cat gpurun_out/r04p/gemm_v8_auto_other_configs.log
