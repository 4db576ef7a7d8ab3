#!/bin/bash
# dev: where should the FAST pass's window sit?  m_ref = sampled maximum + OFFSET leaves 100 + OFFSET log2 units above the sample (a later,
# larger score) and 100 - OFFSET below it (the lane's other query row, whose own maximum is lower): redo fraction on the heavy-tail x8 / x5 weights
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out; log=$out/attn_ref_offset_sweep.log; : > $log
for off in 64 40 24 8; do
  MMPL_EXTRA_HIPCC_FLAGS="-DW64_REF_OFFSET=$off" python -m mmpl_amd.build > /dev/null 2>&1
  for gain in 8 5; do
    python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --heavy-tail --heavy-tail-gain $gain > $out/tmp.json 2>> $out/bench.err
    python - <<PY | tee -a $log
import json; d = json.loads(open("$out/tmp.json").read().strip().splitlines()[-1])
print("W64_REF_OFFSET=$off gain x$gain: step", round(d["sec_per_denoise_step"], 4), "attn", round(d["roofline"]["achieved"], 1), "redo", round(d["attn_blocks_redone_fraction"], 4))
PY
  done
done
python -m mmpl_amd.build > /dev/null 2>&1
