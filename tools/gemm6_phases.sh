#!/bin/bash
# dev: { prologue, k loop, epilogue } shader cycles of gemm_bf16_v6_kernel (timing build on the GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
touch mmpl_amd/csrc/gemm.hip
MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_TIMING=1" python -m mmpl_amd.build 2>&1 | grep -i error | head
timeout 200 python tools/bench_kernels.py gemmphases 2>&1 | grep gemmphases
touch mmpl_amd/csrc/gemm.hip; python -m mmpl_amd.build > /dev/null 2>&1
