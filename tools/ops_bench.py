#!/usr/bin/env python3
"""Measure the operators of SURVEY 8(f) row 3 on the headline graph (Sensor N=1e6, k=8), device
resident, with scipy on the host cores beside them.  One JSON object on stdout.
usage (GPU box): python tools/ops_bench.py [--vertices 1000000] [--nsig 64]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
from scipy.sparse import linalg as sla
from pygsp_amd import engine, graphs

p = argparse.ArgumentParser()
p.add_argument("--vertices", type=int, default=1000000)
p.add_argument("--nsig", type=int, default=64)
p.add_argument("--cg-cols", type=int, default=16)
a = p.parse_args()
ctx = engine.default_context(0)
N, nsig = a.vertices, a.nsig
W, coords = graphs.sensor_weights(N, k=8, seed=42)
perm = engine.locality_order(W, coords)
rng = np.random.default_rng(0)
out = {"graph": "Sensor(N=%d, k=8)" % N, "nnz_W": int(W.nnz), "nsig": nsig}
dw = np.ravel(W.sum(axis=0))
L = sparse.csr_matrix(sparse.diags(dw) - W)
for dtype in (np.float64, np.float32):
    tag = np.dtype(dtype).name
    elt = np.dtype(dtype).itemsize
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
    dev.build_gather_tiles()  # what graphs.Graph does for a graph of this size
    X = rng.standard_normal((N, nsig)).astype(dtype)
    bx, by = ctx.upload(X), ctx.alloc(X.nbytes)
    res = {}
    # ---- L x: bytes = CSR + read x + write y
    ms = min(dev.laplacian_apply_dev(bx.ptr, by.ptr, nsig) for _ in range(5))
    csr = dev.nnz_l * (elt + 4) + 4 * (N + 1)
    res["laplacian_apply"] = {"ms": ms, "alg_GBps": (csr + 2 * X.nbytes) / ms / 1e6,
                              "note": "one launch: the step kernel gathers from x in the caller's order and writes y in it (no permute passes)"}
    y = by.download((N, nsig), dtype)
    t0 = time.perf_counter(); ref = L.dot(X[:, :8].astype(np.float64)); t_cpu = time.perf_counter() - t0
    res["laplacian_apply"]["scipy_1core_ms_per_8cols"] = t_cpu * 1e3
    res["laplacian_apply"]["rel_err"] = float(np.max(np.abs(y[:, :8] - ref)) / np.max(np.abs(ref)))
    # ---- Dirichlet energy of 16 signals
    gram, ms = min((dev.dirichlet_energy_dev(bx.ptr, nsig) for _ in range(3)), key=lambda r: r[1])
    res["dirichlet_energy"] = {"ms": ms, "nsig": nsig}
    x64 = X[:, 0].astype(np.float64)
    res["dirichlet_energy"]["rel_err"] = float(abs(gram[0, 0] - x64.dot(L.dot(x64))) / abs(x64.dot(L.dot(x64))))
    # ---- grad / div
    E = dev.n_edges()
    bg = ctx.alloc(E * nsig * elt)
    ms_g = min(dev.grad_dev(bx.ptr, bg.ptr, nsig) for _ in range(3))
    ms_d = min(dev.div_dev(bg.ptr, by.ptr, nsig) for _ in range(3))
    eb = E * nsig * elt
    res["grad"] = {"ms": ms_g, "n_edges": int(E), "alg_GBps": (eb + X.nbytes + E * (8 + 2 * elt)) / ms_g / 1e6}
    res["div"] = {"ms": ms_d, "alg_GBps": (eb + X.nbytes + E * (4 + 2 * elt) * 2) / ms_d / 1e6}
    z = by.download((N, nsig), dtype)
    res["div"]["rel_err_vs_Lx"] = float(np.max(np.abs(z[:, :8] - ref)) / np.max(np.abs(ref)))
    bg.free()
    # ---- Tikhonov CG
    cols = a.cg_cols
    mask = rng.uniform(size=N) > 0.5
    Y = np.where(mask[:, None], np.sin(6 * coords[:, :1]) + 0.1 * rng.standard_normal((N, cols)), 0.0)
    bm, byy, bxx = ctx.upload(mask.astype(dtype)), ctx.upload(Y.astype(dtype)), ctx.alloc(N * cols * elt)
    tau = 1.0
    iters, ms = dev.tikhonov_cg_dev(tau, bm.ptr, byy.ptr, bxx.ptr, cols)
    xs = bxx.download((N, cols), dtype)
    A = sparse.csr_matrix(sparse.diags(mask.astype(float)) + tau * L)
    t0 = time.perf_counter(); xr, info = sla.cg(A, Y[:, 0].astype(dtype).astype(np.float64)); t_cpu = time.perf_counter() - t0
    res["tikhonov_cg"] = {"columns": cols, "tau": tau, "ms": ms, "iterations": [int(i) for i in iters],
                          "ms_per_iteration": ms / max(int(iters.max()), 1),
                          "scipy_cg_1core_1column_ms": t_cpu * 1e3,
                          "rel_err_col0": float(np.max(np.abs(xs[:, 0] - xr)) / np.max(np.abs(xr)))}
    for b in (bm, byy, bxx, bx, by):
        b.free()
    dev.destroy()
    out[tag] = res
print(json.dumps(out))
