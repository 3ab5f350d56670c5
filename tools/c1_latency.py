#!/usr/bin/env python3
"""BASELINE config 1 (Sensor 1e5, one fp64 signal, order 30: cache resident, latency bound): time of the replayed
call against the lanes per row and rows per wave of the sub-wave kernel.  GPU box only."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    ctx = engine.default_context(0)
    G = graphs.Sensor(100000, seed=42)
    G.estimate_lmax("bounds")
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
    x = np.random.default_rng(0).standard_normal((G.N, 1))
    dev = G.device_graph()
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    rows = []
    for glog2 in (-1, 0, 1, 2, 3, 4, 5):
        for rpw in (0, 1, 2, 4, 8):
            ctx.set_option("narrow_g_log2", glog2)
            ctx.set_option("rows_per_wave", rpw)
            ms = [dev.cheby_filter_dev(c, bx.ptr, by.ptr, 1, float(G.lmax)) for _ in range(25)]
            rows.append({"narrow_g_log2": glog2, "rows_per_wave": rpw, "best_ms": min(ms[3:]), "median_ms": float(np.median(ms[3:]))})
    ctx.set_option("narrow_g_log2", -1)
    ctx.set_option("rows_per_wave", 0)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
