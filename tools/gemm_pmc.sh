#!/bin/bash
# dev: effective clock and MFMA-busy of the GEMM kernels on the block shapes (w64 vs MMPL_GEMM_V6=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in w64 v6; do
  if [ $v = v6 ]; then export MMPL_GEMM_V6=1; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d gpurun_out/gemm_pmc_$v -o p -- python3 tools/bench_kernels.py gemm --iters 3 > gpurun_out/gemm_pmc_$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in ("w64", "v6"):
    for f in glob.glob(f"gpurun_out/gemm_pmc_{v}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter(); seen=set()
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]: continue
            key = (r["Kernel_Name"][:60], r["Grid_Size"])
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); n[key] += 1; dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for key in acc:
            a = acc[key]; cyc = a["GRBM_GUI_ACTIVE"] / 8
            print(v, key, f"n={n[key]} avg_ms={dur[key]/n[key]/1e6:.3f} clock={cyc/dur[key]:.2f}GHz busy={100*a['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc):.1f}% resident={a['SQ_WAVE_CYCLES']*4/1024/cyc:.2f}")
PY
