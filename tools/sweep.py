#!/usr/bin/env python3
"""GPU tuning sweep: time the recurrence-step kernel under different engine options on the
headline workload and print / save GB/s (algorithmic bytes per launch / HIP-event launch time).

usage (on the GPU box): python tools/sweep.py [--n 1000000] [--out gpurun_out/sweep.json]
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsp_amd import engine, filters, graphs  # noqa: E402


def measure(dev, ctx, c, bx, by, nsig, lmax, reps=3):
    best = None
    for _ in range(reps):
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
        t = ctx.last_timing()
        per = t["steps_ms"] / max(t["step_launches"], 1)
        best = per if best is None else min(best, per)
    return best, t


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1000000)
    p.add_argument("--knn", type=int, default=8)
    p.add_argument("--nsig", type=int, default=64)
    p.add_argument("--order", type=int, default=30)
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    p.add_argument("--quick", action="store_true")
    a = p.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    ctx = engine.default_context(0)
    res = {"copy_GBps": ctx.bench_copy(1 << 30, 10), "runs": []}
    print("copy ceiling GB/s:", res["copy_GBps"], flush=True)
    t0 = time.time()
    W, coords = graphs.sensor_weights(a.n, k=a.knn, seed=42)
    print("graph gen s:", time.time() - t0, "nnz_W", W.nnz, flush=True)
    for dtype in (np.float64, np.float32):
        elt = np.dtype(dtype).itemsize
        for reorder in ("morton",):
            perm = engine.locality_order(W, coords) if reorder == "morton" else None
            t0 = time.time()
            dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
            build_s = time.time() - t0
            lmax = 2.0 * float(dev.download_dw().max())
            G = type("G", (), {"lmax": lmax, "e": None})()
            c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=a.order))
            x = np.random.default_rng(0).standard_normal((a.n, a.nsig)).astype(dtype)
            bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
            U = a.n * a.nsig * elt
            csr = dev.nnz_l * (elt + 4) + 4 * (a.n + 1)
            b_launch = (a.order * (csr + 3 * U) + U) / a.order
            vecs = (1, 2) if elt == 8 else (1, 2, 4)
            rpws = (2, 4, 8, 16, 32) if not a.quick else (4,)
            remaps = (1, 0) if reorder == "morton" else (1,)
            kerns = (5, 1)
            for kern, vec, rpw, remap in itertools.product(kerns, vecs, rpws, remaps):
                if kern == 1 and (rpw not in (4,) or vec == 1):
                    continue
                if kern == 5 and vec == 1:
                    continue
                ctx.set_option("kernel", kern)
                ctx.set_option("vec", vec)
                ctx.set_option("rows_per_wave", rpw)
                ctx.set_option("xcd_remap", remap)
                per, t = measure(dev, ctx, c, bx, by, a.nsig, lmax)
                run = {"dtype": np.dtype(dtype).name, "reorder": reorder, "kernel": kern, "vec": vec, "rpw": rpw,
                       "xcd_remap": remap, "ms_per_launch": per, "GBps_alg": b_launch / per / 1e6,
                       "total_ms": t["total_ms"], "permute_ms": t["permute_ms"],
                       "build_s": build_s, "nnz_l": dev.nnz_l, "nnz_int": dev.nnz_internal}
                res["runs"].append(run)
                print(json.dumps(run), flush=True)
                json.dump(res, open(a.out, "w"), indent=1)
            for key, val in (("vec", 0), ("rows_per_wave", 0), ("xcd_remap", 1), ("kernel", 0)):
                ctx.set_option(key, val)
            bx.free(); by.free(); dev.destroy()
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
