#!/usr/bin/env python3
"""Placement tuning on one box: the headline call before tuning, the candidates' launch times, the call after.
    python tools/tune_probe.py [candidates] [f64|f32]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

cands = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
elt = np.dtype(dtype).itemsize
ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42, compute_dtype=dtype)
G.estimate_lmax("bounds")
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
dev = G.device_graph()
x = np.random.default_rng(0).standard_normal((G.N, 64)).astype(dtype)
bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
U = G.N * 64 * elt
b_launch = dev.nnz_l * (elt + 4) + 4 * (G.N + 1) + 3 * U + U / 30


def headline():
    ms = []
    for i in range(10):
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
        t = ctx.last_timing()
        if i >= 3:
            ms.append(t["steps_ms"] / t["step_launches"])
    return round(b_launch / (float(np.median(ms)) * 1e-3) / 8e12, 4)


before = headline()
y0 = by.download(x.shape, dtype)
t0 = time.perf_counter()
rep = dev.tune_placement(c[0], bx.ptr, by.ptr, 64, float(G.lmax), cands, int(os.environ.get("TUNE_STRIDE_MB", "0")))
dt = time.perf_counter() - t0
after = headline()
y1 = by.download(x.shape, dtype)
print(json.dumps({"dtype": np.dtype(dtype).name, "frac_before": before, "frac_after": after, "tuning_s": round(dt, 3),
                  "candidates_launch_ms": [round(v, 4) for v in rep["launch_ms"]], "kept": rep["kept"],
                  "bit_identical": bool(np.array_equal(y0, y1))}))
