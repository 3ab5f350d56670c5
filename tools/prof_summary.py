#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("{:70.70s} calls={:>6} total_ns={:>12} avg_ns={:>10} pct={}".format(
            r.get("Name", ""), r.get("Calls", ""), r.get("TotalDurationNs", ""),
            r.get("AverageNs", ""), r.get("Percentage", "")))
print()
print("== PMC (per kernel: mean counter value per dispatch) ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        acc[k][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0) or 0))
    print("--", os.path.relpath(f, out))
    for k, cs in acc.items():
        if "k_step" not in k and "k_permute" not in k and "k_combine" not in k:
            continue
        for cn, vals in cs.items():
            print("   {:60.60s} {:28s} n={:4d} mean={:.6g}".format(k, cn, len(vals), sum(vals) / len(vals)))
