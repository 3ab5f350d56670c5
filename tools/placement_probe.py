#!/usr/bin/env python3
"""Does the PHYSICAL PLACEMENT of the work panels decide whether a box is "slow" or "fast"?  One process, one GPU: the
headline call measured on a fresh libgspx context (its own workspaces, graph and panels) several times, with device
memory of varying size held or released in between so that the allocator hands out different pages each time.
    python tools/placement_probe.py [trials]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx0 = engine.default_context(0)
coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
W, _, _ = engine.knn_graph(coords, 8, ctx=ctx0)
x = np.random.default_rng(0).standard_normal((1000000, 64))
rng = np.random.default_rng(5)
pads = []
for trial in range(trials):
    streamed = 1 if trial % 2 == 0 else 0
    ctx = engine.Context(0)
    ctx.set_option("streamed_alloc", streamed)
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
    mixt = dev.bench_step_mix(c[0], bx.ptr, by.ptr, 64, float(G.lmax), 1)
    mixt = dev.bench_step_mix(c[0], bx.ptr, by.ptr, 64, float(G.lmax), 1)
    U = G.N * 64 * 8
    b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30
    med = float(np.median(ms))
    print(json.dumps({"trial": trial, "streamed_alloc": streamed, "pads_held_MB": [p.nbytes >> 20 for p in pads],
                      "launch_ms": round(med, 5), "frac": round(b_launch / (med * 1e-3) / 8e12, 4),
                      "mix_frac": round(b_launch / (mixt["steps_ms"] / mixt["step_launches"] * 1e-3) / 8e12, 4),
                      "x_ptr": hex(bx.ptr), "y_ptr": hex(by.ptr)}), flush=True)
    bx.free()
    by.free()
    for g_ in list(G._dev.values()):
        g_.destroy()
    G._dev = {}
    ctx.close()
    # change what the next trial's allocations will land on: hold another pad, or let one go
    if trial % 3 == 2 and pads:
        pads.pop(0).free()
    else:
        pads.append(ctx0.alloc(int(rng.integers(1, 48)) * (64 << 20)))
