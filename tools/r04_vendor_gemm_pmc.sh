#!/bin/bash
# dev (yardstick only, nothing here is on the product path): the vendor library's kernel for the two wide block GEMMs next to
# gemm_bf16_v6_kernel under the same counters -- which kernel it is (name = its tile parameters), matrix-pipe busy x shader clock,
# L2 hit rate, fabric read requests, instruction counts.  usage: tools/r04_vendor_gemm_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export BENCH_SHAPES="qkv:25200:15360:5120:0,ffn0:25200:13824:5120:1,o:25200:5120:5120:3"
for what in gemm gemmref; do
  timeout 300 python tools/bench_kernels.py $what --iters 5 2>&1 | grep -E "^gemm|^vendor" >> $out/wallclock.log
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/pmc_${what}_$i -o p -- python3 tools/bench_kernels.py $what --iters 2 > $out/pmc_${what}_$i.log 2>&1
  done
done
cat $out/wallclock.log
python3 - > $out/summary.txt <<PY
import csv, glob, collections
for what in ("gemm", "gemmref"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])); dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$out/pmc_%s_*/**/*counter_collection.csv" % what, recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            if "gemm_bf16_v6" not in nm and "Cijk" not in nm: continue
            key = (nm[:150], r["Grid_Size"], r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))
            a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("$out/pmc_%s_1/**/*kernel_trace.csv" % what, recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            if "gemm_bf16_v6" not in nm and "Cijk" not in nm: continue
            d = dur[(nm[:150], r["Grid_Size"])]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
    for key, a in sorted(acc.items()):
        c = {k: v[0] / max(v[1], 1) for k, v in a.items()}
        d = dur.get((key[0], key[1]), [0, 0]); ns = d[0] / max(d[1], 1)
        print(what, "| kernel", key[0]); print("    grid", key[1], "wg", key[2], "lds", key[3], "vgpr", key[4], "agpr", key[5], " mean duration under PMC pass 1: %.1f us" % (ns / 1e3))
        if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
            print("    MFMA busy %.1f %%   shader clock %.2f GHz" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"] , c["GRBM_GUI_ACTIVE"] / max(ns, 1)))
        if "TCC_HIT_sum" in c: print("    L2 hit %.1f %%  (hit %.4g miss %.4g)" % (100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), c["TCC_HIT_sum"], c["TCC_MISS_sum"]))
        print("    ", {k: "%.4g" % v for k, v in sorted(c.items())})
PY
cat $out/summary.txt
