#!/usr/bin/env python3
"""One fresh box: the headline call (Sensor 1M, k = 8, 64 fp64 signals, Heat order 30, device resident) in a loop for a
few seconds while rocm-smi is sampled UNDER LOAD - clocks, power, temperatures - next to the measured ms per call.
Looking for what separates the 0.556 boxes from the 0.618 ones (profiles/r05_bench_variance.log).  GPU box only."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402

ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42)
G.estimate_lmax("bounds")
c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=30))
dev = G.device_graph()
x = np.random.default_rng(0).standard_normal((G.N, 64))
bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
samples = []


def sample():
    time.sleep(2.0)
    for _ in range(2):
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse", "--showmemuse"],
                             capture_output=True, text=True).stdout
        keep = [" ".join(ln.split()) for ln in out.splitlines()
                if any(k in ln for k in ("sclk", "mclk", "fclk", "socclk", "Power (W)", "junction", "memory) (C)", "GPU use", "Memory Activity"))]
        samples.append(keep)
        time.sleep(1.0)


th = threading.Thread(target=sample)
th.start()
ms = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 6.0:
    ms.append(dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax)))
th.join()
elt = 8
U = G.N * 64 * elt
b_alg = 30 * (dev.nnz_l * 12 + 4 * (G.N + 1) + 3 * U) + U
med = float(np.median(ms[5:]))
print(json.dumps({"ms_per_call": med, "frac_whole_call": b_alg / (med * 1e-3) / 8e12, "calls": len(ms),
                  "first_calls_ms": ms[:3], "last_calls_ms": ms[-3:], "smi_under_load": samples}))
