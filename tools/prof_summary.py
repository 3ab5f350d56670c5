#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# kernels whose counters are listed (substrings of the kernel name; PROF_KERNELS=a,b,c replaces the default set)
WANTED = [w for w in os.environ.get("PROF_KERNELS", "k_step,k_permute,k_combine").split(",") if w]
STATS_ROWS = int(os.environ.get("PROF_STATS_ROWS", "12"))


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:STATS_ROWS]:
        print("{:70.70s} calls={:>6} total_ns={:>12} avg_ns={:>10} pct={}".format(
            r.get("Name", ""), r.get("Calls", ""), r.get("TotalDurationNs", ""),
            r.get("AverageNs", ""), r.get("Percentage", "")))
# rocprofv3's average is over every launch of the run, warm-up included; bench.py times the launches after
# its warm-up with HIP events.  Same launches, both clocks:
import json as _json
for f in find("stats/**/*kernel_trace.csv"):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
               for r in csv.DictReader(open(f)) if "k_step" in r.get("Kernel_Name", ""))
    try:
        b = _json.loads(open(os.path.join(out, "stats_bench.json")).read().strip().splitlines()[-1])
        timed = int(b["roofline"]["launches_timed"])
        ev = float(b["roofline"]["avg_launch_ms"]) * 1e3
    except Exception:
        continue
    if len(d) >= timed > 0:
        tail = [x[1] for x in d[-timed:]]
        head = [x[1] for x in d[:-timed]]
        print("step-kernel launches: {} in the trace; the last {} (bench.py's timed region): rocprofv3 avg {:.1f} us, "
              "bench.py HIP events of the same launches {:.1f} us; the {} warm-up launches before them avg {:.1f} us".format(
                  len(d), timed, sum(tail) / len(tail) / 1e3, ev, len(head),
                  (sum(head) / len(head) / 1e3) if head else 0.0))
print()
print("== PMC (per kernel: mean counter value per dispatch) ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        acc[k][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0) or 0))
    print("--", os.path.relpath(f, out))
    for k, cs in acc.items():
        if not any(w in k for w in WANTED):
            continue
        for cn, vals in cs.items():
            print("   {:60.60s} {:28s} n={:4d} mean={:.6g}".format(k, cn, len(vals), sum(vals) / len(vals)))

# ---- HBM traffic per recurrence-step launch, for bench.py's roofline.traffic ---------------------
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, i.e. half
# the bytes of a wide coalesced read (MI355X_MICROARCH.md section HBM; re-checked here on
# k_permute_in, a pure copy).  Separate --pmc passes, as the guide prescribes.
import json


def per_kernel(counter):
    res = {}
    for f in find("pmc_%s/**/*counter_collection.csv" % counter):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            res.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return res


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
tot_b, tot_n = 0.0, 0
cal = None
for k in fetch:
    if k not in write:
        continue
    n = min(len(fetch[k]), len(write[k]))
    b = (2.0 * sum(fetch[k][:n]) + sum(write[k][:n])) * 1024.0
    if "k_step" in k:
        tot_b += b
        tot_n += n
    if "k_permute_in" in k:
        cal = {"fetch_KiB": fetch[k][0], "write_KiB": write[k][0]}
if tot_n:
    out_json = {"hbm_bytes_per_launch": tot_b / tot_n, "launches": tot_n,
                "method": "(2*FETCH_SIZE + WRITE_SIZE) KiB per k_step_* dispatch, separate rocprofv3 --pmc passes",
                "copy_kernel_calibration": cal}
    print()
    print("== traffic ==")
    print(json.dumps(out_json))
    json.dump(out_json, open(os.path.join(out, "traffic.json"), "w"))
