#!/usr/bin/env python3
"""Round-4 experiment: two recurrence orders per launch (gspx_cheby_pair_filter_dev) against the default path
on the headline workload (Sensor 1M, k = 8, Heat(50), order 30, 64 signals).  GPU box only.
usage: pair_experiment.py [N] [dtype f64|f32] [nsig] [block_rows,..] [chunk_lanes,..] [ctx option=value ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cheby_oracle as orc  # noqa: E402
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    if os.environ.get("GSPX_PAIR_EXPERIMENT") != "1":
        raise SystemExit("pair_experiment.py: both full-size runs of round 4 (N = 1M, N = 200k) took the GPU box down "
                         "(profiles/r04_pair_experiment.md); set GSPX_PAIR_EXPERIMENT=1 to run it anyway")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
    nsig = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    blocks = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "128").split(",")]
    lanes = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "4").split(",")]
    K = 30
    from pygsp_amd import experimental
    experimental.attach()  # raises on the default library: GSPX_LIB_PATH=pygsp_amd/_lib/libgspx_exp.so
    ctx = engine.default_context(0)
    for kv in sys.argv[6:]:
        key, val = kv.split("=")
        ctx.set_option(key, int(val))
    G = graphs.Sensor(n, k=8, seed=42, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
    dev = G.device_graph()
    elt = np.dtype(dtype).itemsize
    x = np.random.default_rng(0).standard_normal((G.N, nsig)).astype(dtype)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    U = G.N * nsig * elt
    b_alg = K * (dev.nnz_l * (elt + 4) + 4 * (G.N + 1) + 3 * U) + U
    ms = [dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax) for _ in range(8)]
    base = float(np.median(ms[2:]))
    y0 = by.download(x.shape, dtype)
    ref = orc.cheby_op(orc.laplacian(G.W), lmax, c[0], x[:, :1].astype(np.float64))
    out = {"N": G.N, "dtype": np.dtype(dtype).name, "signals": nsig, "order": K, "alg_GB": b_alg / 1e9,
           "default_path": {"ms": base, "frac_8TBs": b_alg / (base * 1e-3) / 8e12,
                            "err_vs_oracle": float(np.max(np.abs(y0[:, 0] - ref)) / np.max(np.abs(ref)))},
           "pair": []}
    for br in blocks:
        t0 = time.perf_counter()
        st = dev.enable_cheb_pair_tiles(br)
        st["build_s"] = time.perf_counter() - t0
        for cw in lanes:
            try:
                ms = [dev.cheby_pair_filter_dev(c[0], bx.ptr, by.ptr, nsig, lmax, chunk_lanes=cw) for _ in range(6)]
            except Exception as e:
                out["pair"].append({"block_rows": br, "chunk_lanes": cw, "error": repr(e)})
                continue
            t = ctx.last_timing()
            best = float(np.median(ms[2:]))
            y = by.download(x.shape, dtype)
            out["pair"].append({"block_rows": br, "chunk_lanes": cw, "ms": best, "frac_8TBs": b_alg / (best * 1e-3) / 8e12,
                                "speedup_vs_default": base / best, "launches": t["step_launches"],
                                "steps_ms": t["steps_ms"], "permute_ms": t["permute_ms"],
                                "max_abs_diff_vs_default": float(np.max(np.abs(y - y0))),
                                "err_vs_oracle": float(np.max(np.abs(y[:, 0] - ref)) / np.max(np.abs(ref))),
                                "tiles": st})
        dev.disable_cheb_pair_tiles()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
