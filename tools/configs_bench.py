#!/usr/bin/env python3
"""Measure every BASELINE.json config that fits one GPU (configs[1..3] + one rank of configs[4]'s
batch), device-resident.  Parity of the same configurations is asserted in tests/test_gpu_1_configs.py and in bench.py's `configs`
(this tool never touches oracle/); here the result is only cross-checked through the
constant-signal identity  p(L) 1 = (c0/2 + sum (-1)^k c_k) 1.  Writes gpurun_out/configs.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsp_amd import engine, filters, graphs  # noqa: E402

ctx = engine.default_context(0)
out = []


def run(name, G, bank, kernels, nsig, K, dtype, check_cols=2, reps=5):
    elt = np.dtype(dtype).itemsize
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    c = np.atleast_2d(np.array(filters.compute_cheby_coeff(bank, m=K)))
    Nf = c.shape[0]
    N = G.N
    dev = G.device_graph()
    x = np.random.default_rng(0).standard_normal((N, nsig)).astype(dtype)
    x[:, 0] = 1.0  # constant column: size-independent self-check (combinatorial Laplacian only)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes * Nf)
    best, tm = 1e9, None
    for _ in range(reps):
        ms = dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
        if ms < best:
            best, tm = ms, ctx.last_timing()
    y = by.download((Nf, N, nsig), dtype)
    err = None
    if G.lap_type == "combinatorial":
        k = np.arange(1, K + 1)
        gain = 0.5 * c[:, 0] + (c[:, 1:] * ((-1.0) ** k)[None, :]).sum(axis=1)
        err = float(np.max(np.abs(y[:, :, 0] - gain[:, None])) / max(np.max(np.abs(gain)), 1e-300))
    U = N * nsig * elt
    csr = dev.nnz_l * (elt + 4) + 4 * (N + 1)
    b_alg = K * (csr + 3 * U) + Nf * U
    r = {"config": name, "dtype": np.dtype(dtype).name, "N": N, "nnz_L": dev.nnz_l, "Nsig": nsig, "Nf": Nf,
         "K": K, "lap": G.lap_type, "total_ms": best, "steps_ms": tm["steps_ms"], "combine_ms": tm["combine_ms"],
         "permute_ms": tm["permute_ms"], "G_units_per_s": N * nsig * K / best / 1e6,
         "B_alg_GB": b_alg / 1e9, "GBps_alg": b_alg / best / 1e6, "frac_8TBps": b_alg / best / 1e6 / 8000,
         "const_signal_identity_err": err, "build_ms": dev.build_ms}
    out.append(r)
    print(json.dumps(r), flush=True)
    bx.free(); by.free()


t0 = time.time()
G = graphs.Sensor(100000, seed=42, compute_dtype=np.float64)
run("C1 Sensor(1e5) comb, Heat K=30, 1 signal", G, filters.Heat(G, 50), None, 1, 30, np.float64, reps=20)
N = 1000000
for dt in (np.float32,):
    for ro in ("auto", "none"):
        G = graphs.ErdosRenyi(N, p=10.0 / N, seed=0, compute_dtype=dt, reorder=ro)
        run("C2 ER(1e6, p=1e-5), MexicanHat x6 K=50, 64 signals [reorder=%s]" % ro, G,
            filters.MexicanHat(G, Nf=6), None, 64, 50, dt, reps=3)
N, k = 2000000, 16
for dt in (np.float64, np.float32):
    for ro in ("auto", "none"):
        G = graphs.StochasticBlockModel(N, k=k, p=9.6e-5, q=2.13e-6, seed=0, lap_type="normalized",
                                        compute_dtype=dt, reorder=ro)
        run("C3 SBM(2e6, k=16) normalized, Heat K=30, 16 signals [reorder=%s]" % ro, G,
            filters.Heat(G, 10), None, 16, 30, dt, reps=3)
for dt in (np.float64, np.float32):
    G = graphs.Sensor(500000, seed=0, compute_dtype=dt)
    run("C4 (one rank) Sensor(5e5), Heat K=30, 32 signals", G, filters.Heat(G, 50), None, 32, 30, dt)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
print("total s", time.time() - t0)
