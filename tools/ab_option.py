#!/usr/bin/env python3
"""A/B of ONE context option on the headline call, alternating in one process on one box:
    python tools/ab_option.py alternate_sweep 0 1 [f64|f32] [nsig]
prints per value the median step-launch time, its fraction of 8 TB/s and a checksum of the result."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

key, values = sys.argv[1], [int(v) for v in sys.argv[2:4]]
dtype = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else np.float64
nsig = int(sys.argv[5]) if len(sys.argv) > 5 else 64
elt = np.dtype(dtype).itemsize
ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42, compute_dtype=dtype)
G.estimate_lmax("bounds")
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
dev = G.device_graph()
x = np.random.default_rng(0).standard_normal((G.N, nsig)).astype(dtype)
bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
U = G.N * nsig * elt
b_launch = dev.nnz_l * (elt + 4) + 4 * (G.N + 1) + 3 * U + U / 30
ms = {v: [] for v in values}
out = {}
for rep in range(6):
    for v in values:
        ctx.set_option(key, v)
        for i in range(4):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, float(G.lmax))
            t = ctx.last_timing()
            if rep and i:
                ms[v].append(t["steps_ms"] / t["step_launches"])
        if rep == 5:
            out[v] = by.download(x.shape, dtype)
res = {}
for v in values:
    med = float(np.median(ms[v]))
    res[str(v)] = {"launch_ms": round(med, 5), "frac": round(b_launch / (med * 1e-3) / 8e12, 4)}
res["max_abs_diff_between_values"] = float(np.max(np.abs(out[values[0]] - out[values[1]])))
res["option"], res["dtype"], res["nsig"] = key, np.dtype(dtype).name, nsig
print(json.dumps(res))
