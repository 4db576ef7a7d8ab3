#!/usr/bin/env python3
"""Post-processes what tools/r02_profiles.sh collected: python3 tools/r02_profiles_post.py <dir> <tag>
   -> <dir>/<tag>_kernel_stats.csv, <dir>/<tag>_pmc_attention_hbm.json, <dir>/<tag>_pmc_mfma_busy.md"""
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

# ---- HBM traffic of the attention op per stage (main launch + split-KV tail + merge of ONE op)
S, d_model = 3600, 5120
stages = {}
for st, (nq, npg) in {"s0": (2, 2), "s1": (7, 9), "s2": (6, 13), "s3": (6, 21)}.items():
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(os.path.join(d, f"hbm_{st}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr and ("attn_w64" in r["Kernel_Name"] or "attn_merge" in r["Kernel_Name"]):
                per_dispatch[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
                names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
        n_ops = max(1, sum(1 for k in names.values() if "attn_w64_kernel<false>" in k))
        vals[ctr] = sum(per_dispatch.values()) / n_ops
    if len(vals) == 2:
        rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
        algo = 2 * (nq * S * d_model * 2) + 2 * (npg * S * d_model * 2)
        stages[st] = {"Lq": nq * S, "Lkv": npg * S, "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                      "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "algorithmic_bytes": algo,
                      "ratio": (rd + wr) / algo}
if stages:
    json.dump({"kernel": "attn_w64_kernel (+ split-KV tail launch + attn_merge_kernel: one attention op)",
               "config": "Wan2.1-T2V-14B 720p, H=40, S=3600",
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/attn_one.py <stage> 2; "
                         "per op = sum over the op's launches; bytes = KiB*1024, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section)",
               "stages": stages}, open(os.path.join(d, f"{tag}_pmc_attention_hbm.json"), "w"), indent=1)

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
print("post-processing done:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(d, f"{tag}_*"))))
