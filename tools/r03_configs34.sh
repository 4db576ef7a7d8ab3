#!/bin/bash
# Functional full-size runs of BASELINE configs[3] / configs[4] on a ONE-GPU box: the real entry point under torch.distributed.run, the
# ranks sharing the GPU over gloo (hand-off staged through the host), synthetic weights.  Wall-clock is NOT a scaling number
# (the ranks share one GPU); the point is that the literally named configurations run end to end at full size.
#   bash tools/r03_configs34.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 - <<'P'
from PIL import Image
import numpy as np
rng = np.random.default_rng(0)
y, x = np.mgrid[0:720, 0:1280]
img = np.stack([(x / 5) % 256, (y / 3) % 256, ((x + y) / 7) % 256], -1) + rng.normal(0, 8, (720, 1280, 3))
Image.fromarray(img.clip(0, 255).astype(np.uint8)).save("/tmp/cond.png")
P
summ() { python3 - "$1" "$2" <<'P'
import sys, torch
v = torch.load(sys.argv[1])
print(sys.argv[2], "video", tuple(v.shape), v.dtype, "mean %.2f std %.2f" % (v.float().mean(), v.float().std()),
      "chunk-to-chunk differs:", bool((v[:81].float() - v[-81:].float()).abs().mean() > 1.0))
P
}
# configs[4] as named, two of its chunks: Wan2.1-I2V-14B model type, 720p, 50 steps, per-chunk image conditioning (one process: two
# 14B / 720p ranks with their 88 GB of KV caches each do not fit one GPU together)
t0=$(date +%s)
timeout 2400 python -m mmpl_amd.cli --synthetic --model 14B --resolution 720p --i2v --i2v_model --image /tmp/cond.png --duration 2 \
  --sampling_steps 50 --output_folder /tmp/out_i2v > $out/cli_i2v_model_14B_720p_2chunks.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - t0 )) s" >> $out/cli_i2v_model_14B_720p_2chunks.log
summ /tmp/out_i2v/0-0.pt "configs[4] (2 chunks, one rank)" >> $out/cli_i2v_model_14B_720p_2chunks.log 2>&1
tail -4 $out/cli_i2v_model_14B_720p_2chunks.log
# configs[3]'s layout at 14B: 4 chunks over 3 ranks (chunk wavefront, hand-off, wrap-around to rank 0, device all-gather), 50 steps;
# 480p so that three 14B ranks fit one GPU
t0=$(date +%s)
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port 29632 -m mmpl_amd.cli --synthetic \
  --model 14B --resolution 480p --duration 4 --sampling_steps 50 --output_folder /tmp/out_t2v --dist_backend gloo > $out/cli_t2v_14B_480p_4chunks_3ranks.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - t0 )) s" >> $out/cli_t2v_14B_480p_4chunks_3ranks.log
summ /tmp/out_t2v/0-0.pt "configs[3] layout (4 chunks, 3 ranks on one GPU)" >> $out/cli_t2v_14B_480p_4chunks_3ranks.log 2>&1
tail -4 $out/cli_t2v_14B_480p_4chunks_3ranks.log
