#!/usr/bin/env python3
"""The headline call in a loop on a FRESH box for a while: ms per call against seconds since the first call, with
rocm-smi sampled along the way - does the step speed change state over time on one GPU (round 6: the same GPU read
0.546 of 8 TB/s seconds after the box started and 0.603 ninety seconds later)?
    python tools/warm_curve.py [seconds] [idle_gap_s]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t_start = time.perf_counter()
ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42)
G.estimate_lmax("bounds")
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
dev = G.device_graph()
x = np.random.default_rng(0).standard_normal((G.N, 64))
bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
U = G.N * 64 * 8
b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = {}
    for ln in out.splitlines():
        for k in ("sclk", "mclk", "fclk", "socclk", "Power (W)", "Sensor memory", "Sensor junction"):
            if k in ln:
                keep[k] = ln.split(":")[-1].strip().strip("()")
    return keep


t0 = time.perf_counter()
rows, next_smi = [], 0.0
print(json.dumps({"setup_s": round(t0 - t_start, 2)}))
while time.perf_counter() - t0 < dur:
    ms = []
    for _ in range(20):
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
        t = ctx.last_timing()
        ms.append(t["steps_ms"] / t["step_launches"])
    now = time.perf_counter() - t0
    row = {"t": round(now, 2), "launch_ms": round(float(np.median(ms)), 5),
           "frac": round(b_launch / (float(np.median(ms)) * 1e-3) / 8e12, 4)}
    if now >= next_smi:
        row["smi"] = smi()
        next_smi = now + 10.0
    rows.append(row)
    print(json.dumps(row), flush=True)
    if gap:
        time.sleep(gap)
