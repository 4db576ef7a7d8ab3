#!/usr/bin/env python3
"""Post-processes what tools/r06_gpu.sh (steps stats / hbm / busy) collected: python3 tools/profiles_post.py <dir> <tag>
   -> <dir>/<tag>_kernel_stats.csv, <dir>/<tag>_pmc_attention_hbm.json, <dir>/<tag>_pmc_hbm_per_kernel.md, <dir>/<tag>_pmc_mfma_busy.md"""
import collections
import csv
import glob
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


# ---- kernel stats
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(d, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# ---- HBM traffic of the self-attention op, measured on the bench process itself: per counter pass, the sum over every
# self-attention launch (main launch + split-KV tail launch + merge) divided by the number of ops (= main launches)
S, d_model = 3600, 5120
vals, n_ops = {}, 0
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(d, f"hbm_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    tot, ops = 0.0, 0
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr or not ("attn_w64" in r["Kernel_Name"] or "attn_merge" in r["Kernel_Name"]):
            continue
        tot += float(r["Counter_Value"])
        if "attn_w64_kernel<false>" in r["Kernel_Name"] and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            ops += 1
    if ops:
        vals[ctr] = tot / ops
        n_ops = ops
if len(vals) == 2:
    rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    shapes = {"s0": (2, 2), "s1": (7, 9), "s2": (6, 13), "s3": (6, 21)}
    algo = sum(2 * (nq * S * d_model * 2) + 2 * (npg * S * d_model * 2) for nq, npg in shapes.values()) / 4.0
    json.dump({"kernel": "attn_w64_kernel (+ split-KV tail launch + attn_merge_kernel: one attention op)",
               "config": "Wan2.1-T2V-14B 720p, H=40, S=3600; the bench rotation s0..s3 (mean over its self-attention ops)",
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 0 "
                         "--no-cpu-baseline --no-vae --no-profile --eager: the bench process itself; per op = sum over all self-attention "
                         "launches / number of ops; bytes = KiB*1024, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section)",
               "ops": n_ops, "FETCH_SIZE_KiB_per_op": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_op": vals["WRITE_SIZE"],
               "hbm_read_bytes_per_op": rd, "hbm_write_bytes_per_op": wr, "mean_hbm_bytes_per_op": rd + wr,
               "algorithmic_bytes_per_op": algo, "ratio": (rd + wr) / algo}, open(os.path.join(d, f"{tag}_pmc_attention_hbm.json"), "w"), indent=1)
    # bench.py reads roofline.traffic from the file THIS names (copy both into profiles/): no sorted glob
    json.dump({"attention_traffic_file": f"{tag}_pmc_attention_hbm.json", "written_by": "tools/profiles_post.py"},
              open(os.path.join(d, "PMC_TRAFFIC.json"), "w"))

# ---- HBM traffic per launch of EVERY kernel (the same two passes): what the memory-bound passes really move
per = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(d, f"hbm_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    tot, seen = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    per[ctr] = {k: (tot[k], len(seen[k])) for k in tot}
if len(per) == 2:
    lines = [f"# {tag} PMC: HBM traffic per launch, every kernel of 4 eager denoise steps of `bench.py` (14B / 720p rotation s0..s3)", "",
             "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); bytes = KiB x 1024, FETCH_SIZE x 2 on gfx950",
             "(MI355X_MICROARCH.md, HBM section).  Mean over the launches of the rotation (stage shapes of 7200 / 25 200 / 21 600 / 21 600 query rows,",
             "mean 18 900): LayerNorm algorithmic = 2 x rows x 5120 x 2 B = 0.387 GB, q / k norm + RoPE + K write = 4 x ... = 0.774 GB (the",
             "cross-attention's q norm, same kernel: 0.387 GB), text cross-attention q + o = 0.387 GB.", "",
             "| kernel | launches | read GB / launch | written GB / launch | total GB / launch |", "|---|---|---|---|---|"]
    ks = sorted(per["FETCH_SIZE"], key=lambda k: -(per["FETCH_SIZE"][k][0] * 2 + per["WRITE_SIZE"].get(k, (0, 1))[0]))
    for k in ks[:24]:
        f, n = per["FETCH_SIZE"][k]
        w, _ = per["WRITE_SIZE"].get(k, (0.0, n))
        rd, wr = f * 1024 * 2 / n / 1e9, w * 1024 / n / 1e9
        lines.append(f"| `{k[:80]}` | {n} | {rd:.3f} | {wr:.3f} | {rd + wr:.3f} |")
    open(os.path.join(d, f"{tag}_pmc_hbm_per_kernel.md"), "w").write("\n".join(lines) + "\n")

# ---- MFMA busy per kernel
fs = glob.glob(os.path.join(d, "busy", "**", "*counter_collection.csv"), recursive=True)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = [f"# {tag} PMC: matrix-pipe occupancy per kernel over 4 denoise steps of `bench.py` (14B / 720p)", "",
             "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 4 --warmup 0 "
             "--no-cpu-baseline --no-vae --no-profile`", "",
             "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of SIMD cycles with",
             "the matrix pipe busy at the clock the kernel actually ran at (clock = GRBM_GUI_ACTIVE / 8 / duration; counter collection",
             "serialises kernels, so durations are not comparable with the timing runs).", "",
             "| kernel | launches | MFMA busy | effective clock GHz | waves resident (SQ_WAVE_CYCLES x 4 / 1024 / cycles) |", "|---|---|---|---|---|"]
    for k in sorted(acc, key=lambda k: -dur[k])[:14]:
        a = acc[k]
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        if cyc <= 0:
            continue
        lines.append(f"| `{k[:70]}` | {cnt[k]} | {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} % | {cyc / dur[k]:.2f} | "
                     f"{a['SQ_WAVE_CYCLES'] * 4 / 1024 / cyc:.2f} |")
    open(os.path.join(d, f"{tag}_pmc_mfma_busy.md"), "w").write("\n".join(lines) + "\n")
# ---- L2 (TCC) hits / misses and fabric reads per launch of the block GEMMs (tools/r06_gpu.sh tcc: bench_kernels.py gemm, ONE shape per pass;
# the pass directory is tcc_<shape>__<counters>; the persistent launch = one block per CU is the product's)
fs = glob.glob(os.path.join(d, "tcc_*__*", "**", "*counter_collection.csv"), recursive=True)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    nlaunch = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in fs:
        shape = re.search(r"tcc_([a-z0-9]+)__", f).group(1)
        for r in csv.DictReader(open(f)):
            if "gemm_" not in r["Kernel_Name"] or int(r["Grid_Size"]) > 512 * 256:      # product launches only: one block per CU
                continue
            key = (shape, short(r["Kernel_Name"])[:44])
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            nlaunch[key][r["Counter_Name"]].add(r["Dispatch_Id"])
    dims = {"qkv": (25200, 15360, 5120), "o": (25200, 5120, 5120), "cq": (25200, 5120, 5120), "co": (25200, 5120, 5120), "ffn0": (25200, 13824, 5120), "ffn2": (25200, 5120, 13824)}
    lines = [f"# {tag} PMC: L2 (TCC) behaviour of the six 14B / 720p block GEMM shapes at M = 25 200, per launch", "",
             "`rocprofv3 --kernel-trace --pmc <TCC_HIT_sum TCC_MISS_sum | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum | FETCH_SIZE>` (separate passes, one shape per pass) "
             "`-- python3 tools/bench_kernels.py gemm --iters 1`; the launches with one block per CU (tile tickets: what `mmpl_dit_forward` issues), mean over them; "
             "FETCH_SIZE bytes = KiB x 1024 x 2 on gfx950.  Operands = (M + N) x K x 2 B; geometry = what 8 XCDs streaming the 4 activation + 8 weight panels of a "
             "4 x 8 tile arrangement once per round of 32 tiles would fetch with perfect sharing inside each L2.", "",
             "| shape (M x N x K) | kernel | launches | L2 hits | L2 misses | hit rate | fabric read requests | fabric read GB | operands GB | geometry GB |", "|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(acc):
        a = acc[key]
        per = lambda c: a[c] / max(len(nlaunch[key][c]), 1) if c in a else float("nan")
        hit, miss = per("TCC_HIT_sum"), per("TCC_MISS_sum")
        M, N, K = dims.get(key[0], (0, 0, 0))
        ops = (M + N) * K * 2 / 1e9
        tiles = -(-M // 256) * -(-N // 256)
        geo = tiles / 32.0 * 12 * 256 * K * 2 / 1e9
        lines.append(f"| {key[0]} {M} x {N} x {K} | `{key[1]}` | {max((len(v) for v in nlaunch[key].values()), default=0)} | {hit:.4g} | {miss:.4g} | "
                     f"{100 * hit / max(hit + miss, 1):.1f} % | {per('TCC_EA0_RDREQ_sum'):.4g} | {per('FETCH_SIZE') * 2048 / 1e9:.3f} | {ops:.3f} | {geo:.2f} |")
    open(os.path.join(d, f"{tag}_pmc_gemm_l2.md"), "w").write("\n".join(lines) + "\n")
print("post-processing done:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(d, f"{tag}_*"))))
