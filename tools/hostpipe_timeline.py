#!/usr/bin/env python3
"""Per-batch host timeline of the pipelined host-array call on the headline workload (gspx_last_host_timeline):
when every batch was packed, shipped, computed, shipped back and unpacked.   python tools/hostpipe_timeline.py [f32]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    dtype = np.float32 if "f32" in sys.argv else np.float64
    N, nsig, K = 1000000, 64, 30
    ctx = engine.default_context(0)
    for kv in [a for a in sys.argv[1:] if "=" in a]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    G = graphs.Sensor(N, k=8, seed=42, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    dev, lmax = G.device_graph(), float(G.lmax)
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
    x = np.random.default_rng(1).standard_normal((N, nsig)).astype(dtype)
    for _ in range(2):
        dev.cheby_filter(c, x, lmax)
    runs = []
    for _ in range(4):
        t0 = time.perf_counter()
        dev.cheby_filter(c, x, lmax)
        wall = (time.perf_counter() - t0) * 1e3
        runs.append({"wall_ms": wall, "stages": ctx.last_host_timing(),
                     "timeline_ms[packed,h2d_issued,kernels_begun,kernels_done,d2h_done,unpacked]":
                         np.round(ctx.last_host_timeline(), 2).tolist()})
    print(json.dumps(min(runs, key=lambda r: r["wall_ms"]), indent=1))


if __name__ == "__main__":
    main()
