#!/usr/bin/env python3
"""Speed class of the work panels against HOW MUCH device memory has been allocated before them: every round holds
another pad, then a new context (its own workspaces: 1.5 GB) runs the headline call.  Maps fast and slow zones of one
card's memory in allocation order (profiles/r06_placement.md).
    python tools/zone_map.py [pad_GB] [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

pad_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 28
ctx0 = engine.default_context(0)
coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
W, _, _ = engine.knn_graph(coords, 8, ctx=ctx0)
x = np.random.default_rng(0).standard_normal((1000000, 64))
bx, by = ctx0.upload(x), ctx0.alloc(x.nbytes)
U = x.nbytes
keep, held_gb = [], 1.0
for r in range(rounds):
    ctx = engine.Context(0)
    G = graphs.Graph(W, coords=coords, ctx=ctx)
    G.estimate_lmax("bounds")
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
    dev = G.device_graph()
    ms = []
    for i in range(8):
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
        t = ctx.last_timing()
        if i >= 2:
            ms.append(t["steps_ms"] / t["step_launches"])
    b_launch = dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U + U / 30
    med = float(np.median(ms))
    # plain stream mixes on buffers allocated at the same depth (gspx_bench_streams allocates its own): do they see the zone?
    st = {k: round(ctx.bench_streams(256 << 20, nr, nw, 0, 8, 3)) for k, (nr, nw) in
          (("r1w1", (1, 1)), ("r3w1", (3, 1)), ("r1w0", (1, 0)), ("r0w1", (0, 1)))} if os.environ.get("ZONE_STREAMS") else {}
    if os.environ.get("ZONE_STREAMS"):  # in place (the recurrence's T_{k-2} -> T_k), 512 MB streams like the headline's panels
        st["r1w1_inplace"] = round(ctx.bench_streams(512 << 20, 1, 1, 4, 8, 3))
        st["r2w1_inplace"] = round(ctx.bench_streams(512 << 20, 2, 1, 4, 8, 3))
        st["r3w2_inplace"] = round(ctx.bench_streams(512 << 20, 3, 2, 4, 8, 3))
        # ... and with the step's walk: 2 workgroups per CU, XCD-partitioned 32 KB blocks
        st["r2w1_inplace_walk"] = round(ctx.bench_streams(512 << 20, 2, 1, 12, 2, 3))
        st["r3w2_inplace_walk"] = round(ctx.bench_streams(512 << 20, 3, 2, 12, 2, 3))
        st["r3w1_walk"] = round(ctx.bench_streams(512 << 20, 3, 1, 8, 2, 3))
        # random gathers of 512-byte rows from a 512 MB panel allocated at this depth (gspx_bench_gather)
        st["gather512"] = round(ctx.bench_gather(1000000, 512, 8000000, 8, 1, 0.0, 8, 3)[1])
        st["gather256"] = round(ctx.bench_gather(2000000, 256, 16000000, 8, 1, 0.0, 8, 3)[1])
    print(json.dumps({"round": r, "GB_allocated_before": round(held_gb, 1), "launch_ms": round(med, 4),
                      "frac": round(b_launch / (med * 1e-3) / 8e12, 4), **st}), flush=True)
    keep.append((ctx, G, dev))
    held_gb += 1.9
    try:
        keep.append(ctx0.alloc(int(pad_gb * (1 << 30))))
        held_gb += pad_gb
    except Exception as e:
        print(json.dumps({"stop": repr(e)}))
        break
