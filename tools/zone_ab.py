#!/usr/bin/env python3
"""What in the step is sensitive to the memory zone?  One process: contexts whose workspaces lie at increasing depths
of the card's memory (8 GB pads in between, tools/zone_map.py's walk); the fastest and the slowest are kept, then the
same option sweeps run on both.  An option under which the two read the same names the mechanism.
    python tools/zone_ab.py [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx0 = engine.default_context(0)
coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
W, _, _ = engine.knn_graph(coords, 8, ctx=ctx0)
x = np.random.default_rng(0).standard_normal((1000000, 64))
bx, by = ctx0.upload(x), ctx0.alloc(x.nbytes)
U = x.nbytes


def launch_ms(ctx, dev, G, c, nsig=64, reps=6, mix=0):
    ms = []
    for i in range(reps):
        if mix:
            t = dev.bench_step_mix(c[0], bx.ptr, by.ptr, nsig, float(G.lmax), mix)
        else:
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, float(G.lmax))
            t = ctx.last_timing()
        if i >= 2:
            ms.append(t["steps_ms"] / t["step_launches"])
    return float(np.median(ms))


cands, pads = [], []
for r in range(rounds):
    ctx = engine.Context(0)
    G = graphs.Graph(W, coords=coords, ctx=ctx)
    G.estimate_lmax("bounds")
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
    dev = G.device_graph()
    cands.append((launch_ms(ctx, dev, G, c), r, ctx, G, dev, c))
    pads.append(ctx0.alloc(10 << 30))
cands.sort(key=lambda t: t[0])
fast, slow = cands[0], cands[-1]
print(json.dumps({"depths_ms": {t[1]: round(t[0], 4) for t in sorted(cands, key=lambda t: t[1])},
                  "fast_round": fast[1], "slow_round": slow[1]}), flush=True)
for t in cands[1:-1]:
    for g_ in list(t[3]._dev.values()):
        g_.destroy()
    t[3]._dev = {}
    t[2].close()
for p in pads:
    p.free()


def both(label, setup=None, nsig=64, mix=0):
    row = {"what": label}
    for name, (_, r, ctx, G, dev, c) in (("fast", fast), ("slow", slow)):
        if setup:
            for k, v in setup.items():
                ctx.set_option(k, v)
        row[name] = round(launch_ms(ctx, dev, G, c, nsig=nsig, mix=mix), 4)
    row["slow_over_fast"] = round(row["slow"] / row["fast"], 3)
    print(json.dumps(row), flush=True)


both("default")


def streams_both(label, nr, nw, nt, wg=8, mb=256):
    row = {"what": label}
    for name, (_, r, ctx, G, dev, c) in (("fast", fast), ("slow", slow)):
        row[name] = round(ctx.bench_streams(mb << 20, nr, nw, nt, wg, 4))
    row["fast_over_slow"] = round(row["fast"] / row["slow"], 3)
    print(json.dumps(row), flush=True)


# plain streams INSIDE each context's own T workspace (gspx_bench_streams bit 4): is it the memory or the step's pattern?
streams_both("streams in the workspace: write only (GB/s)", 0, 1, 16)
streams_both("streams in the workspace: read only", 1, 0, 16)
streams_both("streams in the workspace: 1 read + 1 write", 1, 1, 16)
streams_both("streams in the workspace: 2 reads + 1 write in place", 2, 1, 16 + 4)
streams_both("streams in the workspace: write only, the step's walk, 2 workgroups per CU", 0, 1, 16 + 8, 2)
streams_both("streams in the workspace: 2 writes", 0, 2, 16)
streams_both("fresh allocations: write only", 0, 1, 0)
both("mix kernel (no row products)", mix=1)
both("mix kernel, no barriers", mix=2)
for nt in (0, 1, 2, 4, 8, 5, 10, 15):
    both("tile_nt={}".format(nt), {"tile_nt": nt})
both("tile_nt default", {"tile_nt": -1})
both("alternate_sweep=0", {"alternate_sweep": 0})
both("alternate_sweep=1", {"alternate_sweep": 1})
both("fuse_input=0", {"fuse_input": 0})
both("fuse_input=1", {"fuse_input": 1})
for wg in (256, 384, 448, 512):
    both("tile_workgroups={}".format(wg), {"tile_workgroups": wg})
both("tile_workgroups default", {"tile_workgroups": 0})
both("32 signals (256-byte rows, 256 MB panels)", nsig=32)
both("48 signals", nsig=48)
both("tile_gather=0 (plain gather kernels)", {"tile_gather": 0})
both("tile_gather=1", {"tile_gather": 1})
both("default again")
