#!/bin/bash
# round 4, GPU call 9: measured-wavefront bench lines (bench.py --wavefront-chunks): a 2-chunk video at 14B / 720p, 50 steps, on ONE rank (the
# sequential chain: real pipeline, real hand-off dependency, wall first noise -> last latent), and a 4-chunk video at 1.3B / 480p on two ranks
# sharing the GPU over gloo (functional: the dependency chain across processes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04w; out=gpurun_out/$tag; mkdir -p $out
python bench.py --gpus 2 --dist-backend gloo --model 1.3B --res 480p --wavefront-chunks 4 --sampling-steps 50 > $out/bench_wavefront_1p3B_480p_4chunks_2ranks_gloo_one_gpu.json 2> $out/w1.err; tail -c 1500 $out/bench_wavefront_1p3B_480p_4chunks_2ranks_gloo_one_gpu.json
python bench.py --gpus 1 --wavefront-chunks 2 --sampling-steps 50 > $out/bench_wavefront_14B_720p_2chunks_1rank.json 2> $out/w2.err; tail -c 1500 $out/bench_wavefront_14B_720p_2chunks_1rank.json
tail -3 $out/w1.err $out/w2.err
