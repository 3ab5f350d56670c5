#!/usr/bin/env python3
"""Device k-NN graph construction beside scipy's KD-tree on the host cores.
One JSON object on stdout.  usage (GPU box): python tools/knn_bench.py [--vertices 1000000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse, spatial
from pygsp_amd import engine

p = argparse.ArgumentParser()
p.add_argument("--vertices", type=int, default=1000000)
p.add_argument("--no-host", action="store_true")
a = p.parse_args()
ctx = engine.default_context(0)
N = a.vertices
out = {"N": N, "host_cores": os.cpu_count(), "cases": []}
for d, k in ((2, 8), (2, 6), (3, 10), (1, 4)):
    X = np.random.default_rng(42).uniform(0, 1, (N, d))
    engine.knn_graph(X[:1000], k, ctx=ctx)  # warm-up (module load)
    t0 = time.perf_counter()
    W, sigma, info = engine.knn_graph(X, k, ctx=ctx, neighbors=True)
    t_dev = time.perf_counter() - t0
    case = {"d": d, "k": k, "nnz_W": int(W.nnz), "device_build_ms": info["build_ms"],
            "device_total_s_incl_download": t_dev}
    if not a.no_host:
        t0 = time.perf_counter()
        tree = spatial.cKDTree(X)
        D, NN = tree.query(X, k=k + 1, workers=1)
        t_q1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        D, NN = spatial.cKDTree(X).query(X, k=k + 1, workers=-1)
        t_qall = time.perf_counter() - t0
        t0 = time.perf_counter()
        sg = np.mean(D[:, 1:])
        vals = np.exp(-np.power(D[:, 1:].ravel(), 2) / float(sg))
        Wh = sparse.csc_matrix((vals, (np.repeat(np.arange(N), k), NN[:, 1:].ravel())), shape=(N, N))
        Wh = sparse.csr_matrix((Wh + Wh.T) / 2)
        t_w = time.perf_counter() - t0
        case.update({"host_kdtree_1core_s": t_q1, "host_kdtree_all_cores_s": t_qall,
                     "host_weights_symmetrise_s": t_w,
                     "neighbours_identical": bool(np.array_equal(info["NN"], NN[:, 1:])),
                     "distances_identical": bool(np.array_equal(info["D"], D[:, 1:])),
                     "W_max_rel_diff": float(np.max(np.abs(W.data - Wh.data) / Wh.data)) if W.nnz == Wh.nnz else None,
                     "speedup_vs_1core": (t_q1 + t_w) / (info["build_ms"] * 1e-3)})
    out["cases"].append(case)
print(json.dumps(out))
