#!/bin/bash
# round 4, GPU call 10: after the handshake fix -- the measured wavefront over two ranks (stagger must be the anchor stage, not the chunk)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04w; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_bench_multirank_gpu.py tests/test_wavefront_gpu.py tests/test_cfg_pair_gpu.py -m gpu -q 2>&1 | tail -4 > $out/multirank_tests_after_handshake_fix.log; cat $out/multirank_tests_after_handshake_fix.log
python bench.py --gpus 2 --dist-backend gloo --model 1.3B --res 480p --wavefront-chunks 4 --sampling-steps 50 > $out/bench_wavefront_1p3B_480p_4chunks_2ranks_gloo_one_gpu_after_fix.json 2> $out/w3.err; tail -c 1200 $out/bench_wavefront_1p3B_480p_4chunks_2ranks_gloo_one_gpu_after_fix.json
