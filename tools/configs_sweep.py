#!/usr/bin/env python3
"""BASELINE configs 2 and 3 (graphs without vertex locality): what moves their step time.
For the ER filterbank and the SBM graph, sweep the signals per batch (is a gathered panel that fits the
256 MB Infinity Cache / a 4 MB L2 faster, although the matrix is then streamed once per batch?) and the
plain gather kernels.  Device-resident timing only (parity of the same configs: bench.py, tests/).
Writes gpurun_out/configs_sweep.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsp_amd import engine, filters, graphs  # noqa: E402

ctx = engine.default_context(0)
out = []


def sweep(name, G, bank, nsig, K, dtype, batches, kernels):
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    c = np.atleast_2d(np.array(filters.compute_cheby_coeff(bank, m=K)))
    Nf, N = c.shape[0], G.N
    dev = G.device_graph()
    x = np.random.default_rng(0).standard_normal((N, nsig)).astype(dtype)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes * Nf)
    for kern in kernels:
        for mb in batches:
            ctx.set_option("kernel", kern)
            ctx.set_option("max_batch", mb)
            best, tm = None, None
            for _ in range(3):
                ms = dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
                if best is None or ms < best:
                    best, tm = ms, ctx.last_timing()
            r = {"config": name, "dtype": np.dtype(dtype).name, "kernel": kern, "max_batch": mb, "ms": best,
                 "steps_ms": tm["steps_ms"], "combine_ms": tm["combine_ms"], "launches": tm["step_launches"],
                 "ms_per_launch": tm["steps_ms"] / max(tm["step_launches"], 1)}
            out.append(r)
            print(json.dumps(r), flush=True)
    ctx.set_option("kernel", 0)
    ctx.set_option("max_batch", 0)
    bx.free()
    by.free()


N = 1000000
G = graphs.ErdosRenyi(N, p=10.0 / N, seed=0, compute_dtype=np.float32)
G.estimate_lmax("bounds")  # before the bank is designed
sweep("c2 ER(1e6) MexicanHat x6 K=50 64 signals", G, filters.MexicanHat(G, Nf=6), 64, 50, np.float32,
      (0, 32, 16, 8), (0, 1, 5))
del G
for dt in (np.float64, np.float32):
    G = graphs.StochasticBlockModel(2000000, k=16, p=9.6e-5, q=2.13e-6, seed=0, lap_type="normalized", compute_dtype=dt)
    G.estimate_lmax("bounds")
    sweep("c3 SBM(2e6, k=16) normalized Heat K=30 16 signals", G, filters.Heat(G, 10), 16, 30, dt, (0, 8, 4), (0, 1, 5))
    del G
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs_sweep.json"), "w"), indent=1)
