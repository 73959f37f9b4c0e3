#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv to the encoder's own kernels (m2v::*)
plus one line for everything else (torch kernels that only generate the synthetic clip)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if "m2v::" in r["Name"] or "k_ctl_advance" in r["Name"]]
other = [r for r in rows if r not in keep]
tot = sum(int(r["TotalDurationNs"]) for r in keep) or 1
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfEncoder", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(keep, key=lambda r: -int(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                    "%.2f" % (100.0 * int(r["TotalDurationNs"]) / tot), r["MinNs"], r["MaxNs"], r["StdDev"]])
    w.writerow(["(everything else: torch kernels generating the synthetic clip, copies)",
                sum(int(r["Calls"]) for r in other), sum(int(r["TotalDurationNs"]) for r in other), "", "", "", "", ""])
