#!/usr/bin/env python3
"""HBM traffic per recurrence-step launch of BASELINE configs 2 / 3 from the rocprofv3 --pmc passes of
tools/gpu_prof_configs.sh (gpurun_out/prof_c2, prof_c3): (2 x FETCH_SIZE + WRITE_SIZE) KiB per k_step_* dispatch -
FETCH_SIZE counts 64 bytes per 128-byte request on gfx950 (MI355X_MICROARCH.md; calibrated on the copy kernel of the
same run AND, in round 6, on gathers of unique rows: a row of 64 or 128 bytes costs exactly one 128-byte request -
TCC_EA0_RDREQ_128B = 1.03 per row, four 32-byte DRAM sectors -, a 256-byte row two: profiles/r06_gather_calibration.md,
so the x2 holds for the gather-dominated steps of these configs at every row width) - per compute dtype, next to the
algorithmic bytes of a step.  The figure is FABRIC traffic (L2 misses): Infinity-Cache hits are in it.  Writes profiles/traffic_configs.json, which
bench.py reads into configs[].roofline (PMC counters cannot be read from inside the benchmarked process).
usage: tools/config_traffic.py gpurun_out/prof_c2 gpurun_out/prof_c3"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for d in sys.argv[1:]:
    bench = json.loads(open(os.path.join(d, "stats_bench.json")).read().strip().splitlines()[-1])
    cnt = defaultdict(lambda: defaultdict(list))  # dtype -> counter -> values per dispatch
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_step_" not in k:
                continue
            dt = "f64" if "<double" in k else "f32"
            cnt[dt][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c in bench["configs"]:
        v = cnt.get(c["dtype"])
        if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        hbm = (2 * fetch + write) * 1024
        K = c["order"]
        alg_step = (c["roofline"]["algorithmic_bytes_per_call"] - c["Nf"] * c["N"] * c["Nsig"] * (8 if c["dtype"] == "f64" else 4)) / K
        hit = None
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            h, m = sum(v["TCC_HIT_sum"]), sum(v["TCC_MISS_sum"])
            hit = h / (h + m)
        out["{}_{}".format(c["key"], c["dtype"])] = {
            "hbm_bytes_per_step_launch": hbm, "algorithmic_bytes_per_step": alg_step,
            "traffic_over_algorithmic": hbm / alg_step, "step_launches_counted": len(v["FETCH_SIZE"]),
            "tcc_hit_rate": hit, "avg_step_ms_of_that_run": c["avg_step_ms"],
            "hbm_TBps": hbm / (c["avg_step_ms"] * 1e-3) / 1e12,
            "method": "(2*FETCH_SIZE + WRITE_SIZE) KiB per k_step_* dispatch, separate rocprofv3 --pmc passes of "
                      "`bench.py --no-headline --only-config {}` (tools/gpu_prof_configs.sh)".format(c["key"])}
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_sha: the stamp bench.py compares against)
import datetime
import socket
import subprocess
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
out["_meta"] = {"kernel_sources_sha": bench.kernel_sources_sha(), "git_commit_where_summarised": commit,
                "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"), "host": socket.gethostname(),
                "counter_calibration": "profiles/r06_gather_calibration.md"}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_configs.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
