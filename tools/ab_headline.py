#!/usr/bin/env python3
"""A/B helper: the headline call (Sensor 1M, k = 8, 64 signals, Heat order 30, device resident) in one dtype with the
library GSPX_LIB_PATH points at; prints the median step-launch time and its fraction of 8 TB/s.
    GSPX_LIB_PATH=pygsp_amd/_lib/libgspx_v1.so python tools/ab_headline.py f32 [nsig]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 64
elt = np.dtype(dtype).itemsize
ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42, compute_dtype=dtype)
G.estimate_lmax("bounds")
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
dev = G.device_graph()
x = np.random.default_rng(0).standard_normal((G.N, nsig)).astype(dtype)
bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
ms = []
for i in range(25):
    dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, float(G.lmax))
    t = ctx.last_timing()
    if i >= 5:
        ms.append(t["steps_ms"] / t["step_launches"])
U = G.N * nsig * elt
b_launch = dev.nnz_l * (elt + 4) + 4 * (G.N + 1) + 3 * U + U / 30
med = float(np.median(ms))
print(json.dumps({"lib": os.path.basename(os.environ.get("GSPX_LIB_PATH", "libgspx.so")), "dtype": np.dtype(dtype).name,
                  "nsig": nsig, "launch_ms": round(med, 5), "frac": round(b_launch / (med * 1e-3) / 8e12, 4),
                  "checksum": float(np.abs(by.download(x.shape, dtype)[:1000]).sum())}))
