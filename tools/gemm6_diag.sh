#!/bin/bash
# dev: what bounds gemm_bf16_v6_kernel on the 14B / 720p block shapes -- vendor-library yardstick, L2 hit rate (PMC), and the
# GEMM6_ABL ablation builds (all-L2-hit operands / no epilogue / no DMA).  usage: tools/gemm6_diag.sh <logname> "<abl values>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; : > $out
python -m mmpl_amd.build > /dev/null 2>&1
echo "== shipping build" >> $out
timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out
timeout 300 python tools/bench_kernels.py gemmref --iters 5 2>&1 | grep "^vendor" >> $out
for pmc in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d gpurun_out/gemm6_pmc_$tag -o p -- python3 tools/bench_kernels.py gemm --iters 1 > /dev/null 2>&1
done
python3 - >> $out <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/gemm6_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_v6" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("<")[1][:6] if "<" in r["Kernel_Name"] else "", r["Grid_Size"], r.get("LDS_Block_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])].add(r["Dispatch_Id"])
for key, a in sorted(acc.items()):
    print("pmc", key, {k: f"{v / max(len(n[(key, k)]), 1):.4g}" for k, v in a.items()},
          "L2 hit %.1f%%" % (100 * a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1)) if "TCC_HIT_sum" in a else "")
PY
for abl in $2; do
  echo "== GEMM6_ABL=$abl" >> $out
  MMPL_EXTRA_HIPCC_FLAGS="-DGEMM6_ABL=$abl" python -m mmpl_amd.build > /dev/null 2>&1
  timeout 300 python tools/bench_kernels.py gemm --iters 5 2>&1 | grep "^gemm" >> $out
done
python -m mmpl_amd.build > /dev/null 2>&1
cat $out
