#!/usr/bin/env python3
"""One fresh box: the headline call (Sensor 1M, k = 8, 64 fp64 signals, Heat order 30, device resident) in a loop for a
few seconds while rocm-smi is sampled UNDER LOAD - clocks, power, temperatures - next to the measured ms per call.
Looking for what separates the 0.556 boxes from the 0.618 ones (profiles/r05_bench_variance.log).  Round 6: the MIX
CEILING beside it - the same call with the row products removed from every launch (gspx_bench_step_mix, mode 1 with the
pass barriers, mode 2 without), real and calibration calls alternating - plus the read-only and copy rates of the box:
is a slow box slow for the access mix itself (frac_of_mix_ceiling ~ 1 on both kinds of box) or for the kernel?
GPU box only.  Appends one line to $BOX_PROBE_LOG when set."""
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
# the mix ceiling: real / mode 1 / mode 2 calls alternating (per-launch times from the library's HIP events)
acc = {"real": [], 1: [], 2: []}
for rep in range(9):
    for which in ("real", 1, 2):
        if which == "real":
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, float(G.lmax))
            t = ctx.last_timing()
        else:
            t = dev.bench_step_mix(c[0], bx.ptr, by.ptr, 64, float(G.lmax), which)
        if rep:
            acc[which].append(t["steps_ms"] / t["step_launches"])
mixr = {str(k): float(np.median(v)) for k, v in acc.items()}
b_launch = b_alg / 30
rates = {"read_GBps": ctx.bench_read(1 << 30, 5), "copy_GBps": ctx.bench_copy(1 << 30, 5)}
# what the memory system gives plain stream mixes: reads : writes, 256 MB per stream (gspx_bench_streams)
streams = {}
for nr, nw, nt in ((1, 0, 0), (3, 0, 0), (4, 0, 0), (0, 1, 0), (1, 1, 0), (2, 1, 0), (3, 1, 0), (4, 1, 0), (3, 2, 0),
                   (4, 2, 0), (3, 1, 1), (3, 1, 2), (3, 1, 3)):
    for wg in (8, 2):
        streams["r{}w{}nt{}wg{}".format(nr, nw, nt, wg)] = round(ctx.bench_streams(256 << 20, nr, nw, nt, wg, 4), 1)
rates["streams_GBps"] = streams
try:
    uid = subprocess.run(["rocm-smi", "--showuniqueid", "--showmemvendor", "--showvbios", "--showproductname"],
                         capture_output=True, text=True).stdout
    for key, tag in (("gpu_unique_id", "Unique ID"), ("memory_vendor", "memory vendor"), ("vbios", "VBIOS version"),
                     ("node_id", "Node ID")):
        rates[key] = [ln.split(":")[-1].strip() for ln in uid.splitlines() if ln.startswith("GPU[") and tag in ln][:1]
except Exception:
    pass
rec = {"ms_per_call": med, "frac_whole_call": b_alg / (med * 1e-3) / 8e12, "calls": len(ms),
       "step_launch_ms": mixr["real"], "mix_launch_ms": mixr["1"], "mix_nobarrier_launch_ms": mixr["2"],
       "frac": b_launch / (mixr["real"] * 1e-3) / 8e12, "mix_ceiling_frac": b_launch / (mixr["1"] * 1e-3) / 8e12,
       "mix_nobarrier_ceiling_frac": b_launch / (mixr["2"] * 1e-3) / 8e12,
       "frac_of_mix_ceiling": mixr["1"] / mixr["real"], "frac_of_mix_ceiling_nobarrier": mixr["2"] / mixr["real"],
       **rates, "first_calls_ms": ms[:3], "last_calls_ms": ms[-3:], "smi_under_load": samples}
print(json.dumps(rec))
if os.environ.get("BOX_PROBE_LOG"):
    with open(os.environ["BOX_PROBE_LOG"], "a") as f:
        f.write(json.dumps(rec) + "\n")
