#!/usr/bin/env python3
"""Headline workload (Sensor 1M, k = 8, 64 signals, Heat order 30, device resident) against the context options of
k_step_tile, alternating with the default on one box: is any default stale?  GPU box only.
usage: option_sweep.py [f64|f32] key=v1,v2,... [key=...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
    sweeps = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[2:]]
    ctx = engine.default_context(0)
    G = graphs.Sensor(1000000, k=8, seed=42, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    K = 30
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
    dev = G.device_graph()
    x = np.random.default_rng(0).standard_normal((G.N, 64)).astype(dtype)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)

    def run():
        ms = [dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax)) for _ in range(8)]
        return float(np.median(ms[2:]))
    rows = []
    for key, values in sweeps:
        default = ctx.get_option(key)
        for v in values:
            base = []
            alt = []
            for _ in range(3):  # default / setting / default / setting ...
                ctx.set_option(key, default)
                base.append(run())
                ctx.set_option(key, v)
                alt.append(run())
            ctx.set_option(key, default)
            rows.append({"option": key, "value": v, "default": default, "ms_default": float(np.median(base)),
                         "ms_setting": float(np.median(alt)), "gain_pct": 100 * (np.median(base) / np.median(alt) - 1)})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
