#!/usr/bin/env python3
"""How the relative placement of the work panels moves the headline step: slot-1 / accumulator skews (options ws_skew,
racc_skew) swept on one fresh context per allocator mode; median launch time of the real call and of the mix kernel.
    python tools/skew_sweep.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

ctx0 = engine.default_context(0)
coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
W, _, _ = engine.knn_graph(coords, 8, ctx=ctx0)
x = np.random.default_rng(0).standard_normal((1000000, 64))
SKEWS = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 3 << 19, 2 << 20,
         5 << 19, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 16 << 20]
for streamed in (0, 1):
    ctx = engine.Context(0)
    ctx.set_option("streamed_alloc", streamed)
    ctx.set_option("ws_skew", 32 << 20)   # allocate the workspaces large enough for every skew below, once
    ctx.set_option("racc_skew", 32 << 20)
    G = graphs.Graph(W, coords=coords, ctx=ctx)
    G.estimate_lmax("bounds")
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
    dev = G.device_graph()
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
    U = G.N * 64 * 8
    b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30

    def run(ws, ra):
        ctx.set_option("ws_skew", ws)
        ctx.set_option("racc_skew", ra)
        ms = []
        for i in range(7):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
            t = ctx.last_timing()
            if i >= 2:
                ms.append(t["steps_ms"] / t["step_launches"])
        return round(b_launch / (float(np.median(ms)) * 1e-3) / 8e12, 4)

    base = run(0, 0)
    print(json.dumps({"streamed_alloc": streamed, "skew0": base}), flush=True)
    print(json.dumps({"streamed_alloc": streamed, "ws_skew_sweep_racc0": {s: run(s, 0) for s in SKEWS}}), flush=True)
    print(json.dumps({"streamed_alloc": streamed, "racc_skew_sweep_ws0": {s: run(0, s) for s in SKEWS}}), flush=True)
    print(json.dumps({"streamed_alloc": streamed, "both_equal": {s: run(s, s) for s in SKEWS}}), flush=True)
    print(json.dumps({"streamed_alloc": streamed, "skew0_again": run(0, 0)}), flush=True)
    bx.free()
    by.free()
    for g_ in list(G._dev.values()):
        g_.destroy()
    G._dev = {}
    ctx.close()
