#!/usr/bin/env python3
"""BASELINE config 3 (SBM 2e6, normalized, 16 signals): step time of the plain LDS-staged kernel against the
workgroups resident per CU (option lds_pad_kb) and the rows per wave.  GPU box only."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    ctx = engine.default_context(0)
    rows = []
    for dt in (np.float64, np.float32):
        G = graphs.StochasticBlockModel(2000000, k=16, p=9.6e-5, q=2.13e-6, seed=0, lap_type="normalized", compute_dtype=dt)
        G.estimate_lmax("bounds")
        c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 10), m=30))
        x = np.random.default_rng(0).standard_normal((G.N, 16)).astype(dt)
        dev = G.device_graph()
        bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
        for pad in (0, 8, 16, 29, 40):
            for rpw in (0, 4, 8, 16, 32):
                ctx.set_option("lds_pad_kb", pad)
                ctx.set_option("rows_per_wave", rpw)
                best = min(dev.cheby_filter_dev(c, bx.ptr, by.ptr, 16, 2.0) for _ in range(3))
                tm = ctx.last_timing()
                rows.append({"dtype": np.dtype(dt).name, "lds_pad_kb": pad, "rows_per_wave": rpw, "ms": best,
                             "step_ms": tm["steps_ms"] / tm["step_launches"]})
        ctx.set_option("lds_pad_kb", 0)
        ctx.set_option("rows_per_wave", 0)
        bx.free()
        by.free()
        for g_ in list(G._dev.values()):
            g_.destroy()
        G._dev = {}
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
