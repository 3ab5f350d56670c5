#!/usr/bin/env python3
"""counted / true per row width from the passes of tools/gpu_gather_calibration.sh (markdown on stdout)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
plain = json.loads(open(os.path.join(d, "plain.json")).read().strip().splitlines()[-1])
true = {l["lanes_per_row"]: l for l in plain["launches"]}
cnt = defaultdict(lambda: defaultdict(list))  # lanes per row -> counter -> per-dispatch values
copy = defaultdict(list)
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        m = re.search(r"k_bench_gather<(\d+),\s*(\d+)>", k)
        if m:
            cnt[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "k_permute_in" in k:
            copy[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# Fabric counters on gathers of UNIQUE rows (k_bench_gather, {} rows per launch, every row once)".format(plain["rows"]))
print()
print("True bytes of a launch: rows x row bytes of payload + 4 bytes of index per row (read coalesced).")
print()
names = sorted({c for v in cnt.values() for c in v})
print("| row bytes | payload MB | ms | rows/us | " + " | ".join(names) + " |")
print("|---|---|---|---|" + "---|" * len(names))
for lpr in sorted(cnt):
    t = true[lpr]
    cells = []
    for c in names:
        v = cnt[lpr].get(c)
        cells.append("{:.6g}".format(sum(v) / len(v)) if v else "-")
    print("| {} | {:.1f} | {:.3f} | {:.1f} | ".format(t["row_bytes"], t["payload_bytes"] / 1e6, t["ms"], t["rows_per_us"]) + " | ".join(cells) + " |")
print()
print("## counted / true")
print()
print("| row bytes | FETCH_SIZE KiB x 1024 / payload | per row (bytes) | 2 x that | TCC_EA0_RDREQ per row | 32B requests per row | TCC_MISS per row | TCC_HIT per row |")
print("|---|---|---|---|---|---|---|---|")
for lpr in sorted(cnt):
    t = true[lpr]
    rows = t["rows"]

    def per_row(name):
        v = cnt[lpr].get(name)
        return (sum(v) / len(v) / rows) if v else None
    fs = per_row("FETCH_SIZE")
    fmt = lambda x, s="{:.3f}": s.format(x) if x is not None else "-"
    print("| {} | {} | {} | {} | {} | {} | {} | {} |".format(
        t["row_bytes"], fmt(fs * 1024 * rows / t["payload_bytes"] if fs else None), fmt(fs * 1024 if fs else None, "{:.1f}"),
        fmt(fs * 2048 if fs else None, "{:.1f}"), fmt(per_row("TCC_EA0_RDREQ_sum")), fmt(per_row("TCC_EA0_RDREQ_32B_sum")),
        fmt(per_row("TCC_MISS_sum")), fmt(per_row("TCC_HIT_sum"))))
print()
if copy:
    print("Copy kernel of the same runs (k_permute_in, 512 MiB each way): " +
          ", ".join("{} = {:.6g}".format(k, sum(v) / len(v)) for k, v in sorted(copy.items())))
