#!/bin/bash
# round 4, GPU call 4: v8 with the L2 share prefetch (bit-identity, placement sweep, in-situ A/B of the per-shape choice), the reference
# algorithm's executor floor over the 408-forward trajectory, the fused-norm test against the oracle, the wavefront test repeated, full suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r04d; out=gpurun_out/$tag; mkdir -p $out
MMPL_GEMM_V8=0 timeout 600 python tools/gemm_v8_check.py > $out/gemm_v6_sha.log 2>&1
MMPL_GEMM_V8=1 timeout 600 python tools/gemm_v8_check.py > $out/gemm_v8_sha.log 2>&1
diff $out/gemm_v6_sha.log $out/gemm_v8_sha.log > $out/gemm_v8_vs_v6.diff && echo "v8 == v6 bit for bit" | tee -a $out/gemm_v8_vs_v6.diff; head -20 $out/gemm_v8_vs_v6.diff
for flags in "" "-DGEMM8_PF=0" "-DGEMM8_DMAS=2 -DGEMM8_BAR2=24" "-DGEMM8_DMAS=2 -DGEMM8_BAR2=30" "-DGEMM8_BAR1=32"; do
  echo "== v8 build flags: [$flags]" >> $out/gemm_v8_sweep.log
  if [ -n "$flags" ]; then MMPL_EXTRA_HIPCC_FLAGS="$flags" python -m mmpl_amd.build > /dev/null 2>&1; fi
  MMPL_GEMM_V8=1 timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $out/gemm_v8_sweep.log
done
python -m mmpl_amd.build > /dev/null 2>&1
echo "== v6 (MMPL_GEMM_V8=0)" >> $out/gemm_v8_sweep.log
MMPL_GEMM_V8=0 timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" | sed 's/|  + split.*//' >> $out/gemm_v8_sweep.log
cat $out/gemm_v8_sweep.log
for v in 0 auto 1 0 auto; do
  if [ $v = auto ]; then unset MMPL_GEMM_V8; else export MMPL_GEMM_V8=$v; fi
  python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-vae --profile-all > $out/bench_tmp.json 2>> $out/bench.err
  python - <<PY | tee -a $out/bench_v8_insitu_ab.log
import json; d = json.loads(open("$out/bench_tmp.json").read().strip().splitlines()[-1])
print("V8=$v step", round(d["sec_per_denoise_step"], 4), [round(x, 4) for x in d["sec_per_denoise_step_by_stage"]], "gemm", round(d.get("gemm_tflops"), 1), "attn", round(d["roofline"]["achieved"], 1))
PY
done
unset MMPL_GEMM_V8
timeout 900 python tools/traj_executor_floor.py > $out/traj_executor_floor.log 2>&1; tail -2 $out/traj_executor_floor.log
python -m pytest tests/test_vae_gpu.py -m gpu -q -s 2>&1 | grep -E "rel_l2|passed|failed" > $out/vae_tests.log; cat $out/vae_tests.log
for i in 1 2 3; do python -m pytest tests/test_wavefront_gpu.py -m gpu -q -k real_pipeline 2>&1 | grep -E "differ|passed|failed" >> $out/wavefront_repeat.log; done; cat $out/wavefront_repeat.log
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -6 $out/gputests.log
ls $out
