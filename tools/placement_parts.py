#!/usr/bin/env python3
"""WHICH allocation's physical placement moves the headline step?  One process, one GPU; every phase re-creates one
group of allocations `trials` times (holding the previous ones so that other pages are drawn) while the rest stays:
  ws     the streamed workspaces (T slots, accumulator) - gspx_ctx_tune_placement's candidates
  graph  the device graph (factor values, tile positions, row lists, headers, CSR)
  xy     the caller's input and output panels
  all    everything (what a fresh process draws)
    python tools/placement_parts.py [trials]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = engine.default_context(0)
coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
W, _, _ = engine.knn_graph(coords, 8, ctx=ctx)
x = np.random.default_rng(0).standard_normal((1000000, 64))
U = x.nbytes


def make_graph():
    G = graphs.Graph(W, coords=coords, ctx=ctx)
    G.estimate_lmax("bounds")
    return G, G.device_graph()


def frac(dev, G, c, bx, by):
    ms = []
    for i in range(9):
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
        t = ctx.last_timing()
        if i >= 3:
            ms.append(t["steps_ms"] / t["step_launches"])
    b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30
    return round(b_launch / (float(np.median(ms)) * 1e-3) / 8e12, 4)


G, dev = make_graph()
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
bx, by = ctx.upload(x), ctx.alloc(U)
out = {"start": frac(dev, G, c, bx, by)}
rep = dev.tune_placement(c[0], bx.ptr, by.ptr, 64, float(G.lmax), trials)
b7 = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30
out["ws"] = {"candidates_frac_short_call": [round(b7 / (v * 1e-3) / 8e12, 4) for v in rep["launch_ms"]], "kept": rep["kept"],
             "after": frac(dev, G, c, bx, by)}
print(json.dumps(out), flush=True)
keep = []
res = []
for t in range(trials):  # the caller's panels
    keep += [bx, by]
    bx, by = ctx.upload(x), ctx.alloc(U)
    res.append(frac(dev, G, c, bx, by))
print(json.dumps({"xy": res}), flush=True)
res = []
for t in range(trials):  # the graph
    keep.append((G, dev))
    G, dev = make_graph()
    res.append(frac(dev, G, c, bx, by))
print(json.dumps({"graph": res}), flush=True)
res = []
for t in range(trials):  # everything, on a fresh context each (its own workspaces)
    keep += [bx, by, (G, dev)]
    c2 = engine.Context(0)
    G2 = graphs.Graph(W, coords=coords, ctx=c2)
    G2.estimate_lmax("bounds")
    d2 = G2.device_graph()
    bx2, by2 = c2.upload(x), c2.alloc(U)
    ms = []
    for i in range(9):
        d2.cheby_filter_dev(c, bx2.ptr, by2.ptr, 64, float(G2.lmax))
        tt = c2.last_timing()
        if i >= 3:
            ms.append(tt["steps_ms"] / tt["step_launches"])
    b_launch = d2.nnz_l * 12 + 4 * (G2.N + 1) + 3 * U + U / 30
    res.append(round(b_launch / (float(np.median(ms)) * 1e-3) / 8e12, 4))
    keep += [bx2, by2, (G2, d2), c2]
print(json.dumps({"all_fresh_context": res}), flush=True)
