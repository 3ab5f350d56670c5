#!/usr/bin/env python3
"""Soak of the LDS-staged recurrence kernel against the oracle: many panel widths x dtypes x orders on
Hilbert-ordered sensor graphs of several sizes, repeated; plus the randomised differential test of
tests/test_gpu_9_fuzz.py on extra seeds.  Looks for rare events (races between waves), not for speed.
GPU box only.  usage: tools/soak.py [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cheby_oracle as orc  # noqa: E402
from pygsp_amd import engine, graphs  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = engine.default_context(0)
rng = np.random.default_rng(0)
bad = 0
runs = 0
for N in (40000, 200000, 700000):
    W, coords = graphs.sensor_weights(N, k=8, seed=N)
    L = orc.laplacian(W)
    lmax = 2.0 * float(np.ravel(W.sum(0)).max())
    for dtype in (np.float64, np.float32):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=engine.locality_order(W, coords), ctx=ctx)
        assert dev.build_gather_tiles()["slow_blocks"] == 0
        tol = 1e-11 if dtype == np.float64 else 3e-5
        for nsig in (4, 8, 12, 16, 20, 32, 36, 64, 72, 96, 128, 160):
            x = rng.standard_normal((N, nsig))
            for order in (1, 2, 3, 5, 11, 30):
                c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, order)
                cols = [0, nsig // 2, nsig - 1]
                ref = orc.cheby_op(L, lmax, c, x[:, cols].astype(dtype).astype(np.float64))
                for _ in range(rounds):
                    y, _ms = dev.cheby_filter(c, x, lmax)
                    err = float(np.max(np.abs(y[0][:, cols] - ref)) / np.max(np.abs(ref)))
                    runs += 1
                    if not err < tol:
                        bad += 1
                        print("BAD", N, np.dtype(dtype).name, nsig, order, err, flush=True)
        dev.destroy()
    print("N", N, "done; runs", runs, "bad", bad, flush=True)
import test_gpu_9_fuzz as fuzz  # noqa: E402
for seed in range(4, 16):
    try:
        fuzz.test_fuzz_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FUZZ BAD seed", seed, str(e)[:300], flush=True)
print("soak finished: runs", runs, "+ 12 fuzz seeds, bad", bad)
sys.exit(1 if bad else 0)
