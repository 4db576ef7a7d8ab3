#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports:  python tools/rocpd_summary.py <results.db> <out.csv> [--md out.md]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    return name


def main():
    db, out = sys.argv[1], sys.argv[2]
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "LDS", "Scratch",
           "Workgroup"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(hdr)
        for r in rows:
            w.writerow([short(r[0]), r[1], int(r[2]), round(r[3], 1), round(100.0 * r[2] / total, 3), int(r[4]), int(r[5]), *r[6:]])
    if md:
        with open(md, "w") as f:
            f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in rows[:25]:
                f.write(f"| `{short(r[0])}` | {r[1]} | {r[2] / 1e6:.2f} | {r[3] / 1e3:.1f} | {100.0 * r[2] / total:.2f} |\n")
    print(f"{len(rows)} kernels, total {total / 1e9:.3f} s -> {out}")


if __name__ == "__main__":
    main()
