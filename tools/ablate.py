#!/usr/bin/env python3
"""Ablation on synthetic matrices: where does the recurrence step spend its time?
  diag   : L = diagonal only (1 real entry per row)      -> streaming floor (3 vectors)
  band   : 11-entry band (perfect gather locality)        -> + cached gathers
  sensor : kNN sensor graph, Morton order                 -> + real locality
  sensor_rand : same graph, random vertex order           -> no locality
"""
import json
import os
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsp_amd import engine, graphs  # noqa: E402

N, NSIG, K = 1000000, 64, 12
ctx = engine.default_context(0)
out = []


def band(n, half):
    offs = [o for o in range(-half, half + 1)]
    diags = [np.full(n - abs(o), 1.0 + 0.01 * abs(o)) for o in offs]
    return sparse.diags(diags, offs, shape=(n, n), format="csr")


W, coords = graphs.sensor_weights(N, k=8, seed=42)
Ls = orc = None
mats = {
    "diag": (sparse.identity(N, format="csr") * 2.0, None),
    "band11": (band(N, 5), None),
    "sensor_morton": (None, engine.locality_order(W, coords)),
    "sensor_rand": (None, None),
}
c = np.atleast_2d(np.linspace(1.0, 0.1, K + 1))
for dtype in (np.float64, np.float32):
    x = np.random.default_rng(0).standard_normal((N, NSIG)).astype(dtype)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    for name, (L, perm) in mats.items():
        if L is None:
            dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        else:
            dev = engine.DeviceGraph.from_l(L, dtype=dtype, ctx=ctx)
        for kern, vec in ((1, 0), (5, 0)):
            ctx.set_option("kernel", kern)
            ctx.set_option("vec", vec)
            best = 1e9
            for _ in range(3):
                dev.cheby_filter_dev(c, bx.ptr, by.ptr, NSIG, 4.0)
                t = ctx.last_timing()
                best = min(best, t["steps_ms"] / t["step_launches"])
            r = {"dtype": np.dtype(dtype).name, "matrix": name, "kernel": kern, "ms_per_launch": best,
                 "nnz_int": dev.nnz_internal, "nnz_l": dev.nnz_l}
            out.append(r)
            print(json.dumps(r), flush=True)
        dev.destroy()
    bx.free(); by.free()
ctx.set_option("kernel", 0)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ablate.json"), "w"), indent=1)
