#!/bin/bash
# round 4, GPU call 7: the allocator-placement regression test; the 256x128 kernel (v2) on the Wan 1.3B / 480p short-K shapes next to v6 + sub-tile
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04g; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_dit_forward_gpu.py -m gpu -q -k "workspace_lies or golden_480p" 2>&1 | tail -3 > $out/placement_test.log; cat $out/placement_test.log
export BENCH_SHAPES="qkv_s1:10920:4608:1536:0,o_s1:10920:1536:1536:3,ffn0_s1:10920:8960:1536:1,ffn2_s1:10920:1536:8960:3,qkv_s0:3120:4608:1536:0,o_s0:3120:1536:1536:3,ffn0_s0:3120:8960:1536:1,ffn2_s0:3120:1536:8960:3,qkv_s2:9360:4608:1536:0,o_s2:9360:1536:1536:3,ffn2_s2:9360:1536:8960:3"
for v in 0 1 0 1; do
  echo "== MMPL_GEMM_V2=$v" >> $out/gemm_v2_1p3B_shapes.log
  MMPL_GEMM_V2=$v timeout 300 python tools/bench_kernels.py gemm --iters 20 2>&1 | grep "^gemm" >> $out/gemm_v2_1p3B_shapes.log
done
cat $out/gemm_v2_1p3B_shapes.log
