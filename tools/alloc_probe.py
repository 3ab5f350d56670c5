#!/usr/bin/env python3
"""What the physical backing of the work panels does to the headline step, on one GPU in one process: plain hipMalloc,
2 MB chunks scrambled (the default) / in order, larger chunks, one physical piece per workspace - a fresh context per
mode, two rounds in alternating order.
    python tools/alloc_probe.py"""
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
MODES = [("plain hipMalloc", {"streamed_alloc": 0}),
         ("2 MB scrambled (default)", {"streamed_alloc": 1, "streamed_chunk_mb": 2, "streamed_scramble": 1}),
         ("2 MB in order", {"streamed_alloc": 1, "streamed_chunk_mb": 2, "streamed_scramble": 0}),
         ("32 MB scrambled", {"streamed_alloc": 1, "streamed_chunk_mb": 32, "streamed_scramble": 1}),
         ("256 MB scrambled", {"streamed_alloc": 1, "streamed_chunk_mb": 256, "streamed_scramble": 1}),
         ("one piece", {"streamed_alloc": 1, "streamed_chunk_mb": 0, "streamed_scramble": 0})]
for rnd in range(2):
    for name, opts in (MODES if rnd == 0 else MODES[::-1]):
        ctx = engine.Context(0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        G = graphs.Graph(W, coords=coords, ctx=ctx)
        G.estimate_lmax("bounds")
        c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
        dev = G.device_graph()
        bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
        ms = []
        for i in range(12):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
            t = ctx.last_timing()
            if i >= 3:
                ms.append(t["steps_ms"] / t["step_launches"])
        U = G.N * 64 * 8
        b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30
        med = float(np.median(ms))
        print(json.dumps({"round": rnd, "mode": name, "launch_ms": round(med, 5),
                          "frac": round(b_launch / (med * 1e-3) / 8e12, 4)}), flush=True)
        bx.free()
        by.free()
        for g_ in list(G._dev.values()):
            g_.destroy()
        G._dev = {}
        ctx.close()
