#!/usr/bin/env python3
"""End-to-end time of the host-pointer call (numpy in, numpy out) on the headline workload, over the pipeline's
batch width and thread count; one-shot form beside it.   python tools/hostpipe_sweep.py [f64|f32] > out.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    dtype = np.float32 if "f32" in sys.argv else np.float64
    N, nsig, K = 1000000, 64, 30
    ctx = engine.default_context(0)
    coords = np.random.default_rng(42).uniform(0, 1, (N, 2))
    W, _, _ = engine.knn_graph(coords, 8, ctx=ctx)
    G = graphs.Graph(W, coords=coords, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    dev, lmax = G.device_graph(), float(G.lmax)
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
    x = np.random.default_rng(1).standard_normal((N, nsig)).astype(dtype)
    rows = []

    def timed(label):
        dev.cheby_filter(c, x, lmax)
        best, st, kms = None, None, None
        for _ in range(3):
            t0 = time.perf_counter()
            y, ms = dev.cheby_filter(c, x, lmax)
            dt = (time.perf_counter() - t0) * 1e3
            if best is None or dt < best:
                best, st, kms = dt, ctx.last_host_timing(), ms
        rows.append({"config": label, "ms": best, "kernel_ms": kms, "stages": st})
        return y

    ctx.set_option("host_pipeline", 0)
    y0 = timed("one-shot")
    ctx.set_option("host_pipeline", 2)
    for threads in (0, 8):  # host_batch 0: the engine's own schedule (half-width first and last batch)
        ctx.set_option("host_batch", 0)
        ctx.set_option("host_edge", 0)
        ctx.set_option("host_threads", threads)
        y = timed("auto schedule, threads {}".format(threads or "auto"))
        rows[-1]["identical_to_one_shot"] = bool(np.array_equal(y, y0))
    elt = np.dtype(dtype).itemsize
    # (batch, first / last batch) in bytes per row
    for bb, eb in ((32, 32), (64, 32), (64, 64), (96, 32), (96, 64), (128, 32), (128, 64), (128, 128), (192, 64), (256, 128)):
        batch, edge = bb // elt, eb // elt
        ctx.set_option("host_batch", batch)
        ctx.set_option("host_edge", edge)
        ctx.set_option("host_threads", 16)
        y = timed("batch {} edge {} threads 16".format(batch, edge))
        rows[-1]["identical_to_one_shot"] = bool(np.array_equal(y, y0))
    ctx.set_option("host_edge", 0)
    for batch in ((16,) if "auto" in sys.argv else (8, 16, 32)):
        for threads in ((16,) if "auto" in sys.argv else (2, 4, 8, 16)):
            ctx.set_option("host_batch", batch)
            ctx.set_option("host_threads", threads)
            y = timed("batch {} threads {}".format(batch, threads))
            rows[-1]["identical_to_one_shot"] = bool(np.array_equal(y, y0))
    print(json.dumps({"dtype": np.dtype(dtype).name, "N": N, "Nsig": nsig, "order": K, "host_cores": os.cpu_count(),
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
