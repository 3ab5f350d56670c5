#!/usr/bin/env python3
"""k-NN graph construction beyond three dimensions (tiled brute force, pair distances on MFMA) beside scipy's
KD-tree on all host cores.  One JSON object on stdout.  GPU box only."""
import ctypes
import json
import os
import sys
import time

import numpy as np
from scipy import spatial

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import _capi, engine  # noqa: E402


def main():
    ctx = engine.default_context(0)
    lib = _capi.load()
    out = {"host_cores": os.cpu_count(), "cases": []}
    engine.knn_graph(np.random.default_rng(0).standard_normal((2000, 8)), 5, ctx=ctx)  # warm-up
    for N, d, k, host_rows in ((50000, 9, 8, 50000), (100000, 16, 10, 20000), (100000, 25, 10, 5000),
                               (200000, 8, 10, 50000), (50000, 64, 16, 2000)):
        X = np.random.default_rng(N + d).standard_normal((N, d))
        ctx.set_option("knn_f32", 0)  # the candidate sweep on the fp64 matrix cores ...
        _, _, info64 = engine.knn_graph(X, k, ctx=ctx, neighbors=True)
        ctx.set_option("knn_f32", 1)  # ... and the default: on the fp32 ones where their rounding margin allows
        t0 = time.perf_counter()
        W, sigma, info = engine.knn_graph(X, k, ctx=ctx, neighbors=True)
        t_dev = time.perf_counter() - t0
        h = ctypes.c_void_p()
        _capi.check(lib.gspx_knn_build(ctx._h, N, d, _capi.ptr(X), k, 0.0, 0, 0, ctypes.byref(h)))
        st = np.zeros(4)
        _capi.check(lib.gspx_knn_search_stats(h, _capi.ptr(st)))
        lib.gspx_knn_destroy(h)
        pairs = float(N) * N
        case = {"N": N, "d": d, "k": k, "device_build_ms": info["build_ms"], "device_total_s_incl_download": t_dev,
                "device_build_ms_fp64_sweep": info64["build_ms"],
                "same_result_either_sweep": bool(np.array_equal(info["NN"], info64["NN"]) and np.array_equal(info["D"], info64["D"])),
                "pair_distances_per_s": pairs / (info["build_ms"] * 1e-3),
                "mfma_flops_per_s": pairs * 2 * (4 * (4 if d <= 16 else 8 if d <= 32 else 16)) / (info["build_ms"] * 1e-3),
                "sample": int(st[0]), "capacity": int(st[1]), "mean_candidates": st[2], "exact_scans": int(st[3])}
        rows = np.arange(host_rows)
        t0 = time.perf_counter()
        D, NN = spatial.cKDTree(X).query(X[rows], k=k + 1, workers=-1)
        t_host = time.perf_counter() - t0
        case.update({"host_kdtree_all_cores_s_for_rows": t_host, "host_rows": host_rows,
                     "host_s_extrapolated_to_N": t_host * N / host_rows,
                     "neighbours_identical": bool(np.array_equal(info["NN"][rows], NN[:, 1:])),
                     "distances_identical": bool(np.array_equal(info["D"][rows], D[:, 1:]))})
        out["cases"].append(case)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
