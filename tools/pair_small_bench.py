#!/usr/bin/env python3
"""BASELINE config 1 (Sensor(100000), one fp64 signal, Heat, order 30) and its neighbours: the replayed call with
option pair_small = 0 / 1 (single steps vs two orders per launch).  GPU box only; the option exists in the EXPERIMENTAL
build only (make -C pygsp_amd/csrc experimental; GSPX_LIB_PATH=pygsp_amd/_lib/libgspx_exp.so)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cheby_oracle as orc  # noqa: E402
from pygsp_amd import engine, filters, graphs  # noqa: E402

from pygsp_amd import experimental  # noqa: E402
experimental.attach()  # raises on the default library
ctx = engine.default_context(0)
rows = []
for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '50000,100000,200000').split(',')]:
    for dtype in (np.float64, np.float32):
        sys.stderr.write('graph %d %s\n' % (n, np.dtype(dtype).name))
        sys.stderr.flush()
        G = graphs.Sensor(n, seed=42, compute_dtype=dtype)
        G.estimate_lmax("bounds")
        lmax = float(G.lmax)
        c = filters.compute_cheby_coeff(filters.Heat(G, 50), m=30)
        dev = G.device_graph()
        for nsig in (1, 2):
            x = np.random.default_rng(0).standard_normal((G.N, nsig)).astype(dtype)
            bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
            ref = orc.cheby_op(orc.laplacian(G.W), lmax, c, x[:, :1].astype(np.float64))
            r = {"N": n, "dtype": np.dtype(dtype).name, "signals": nsig}
            for mode in (0, 1, 0, 1):
                ctx.set_option("pair_small", mode)
                ms = [dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax) for _ in range(24)]
                y = by.download(x.shape, dtype)
                best = float(np.median(ms[5:]))
                r["ms_pair" if mode else "ms_single"] = min(best, r.get("ms_pair" if mode else "ms_single", 1e9))
                r["err_pair" if mode else "err_single"] = float(np.max(np.abs(y[:, 0] - ref[:, 0])) / np.max(np.abs(ref)))
            r["speedup"] = r["ms_single"] / r["ms_pair"]
            rows.append(r)
            sys.stderr.write(json.dumps(r) + '\n')
            sys.stderr.flush()
            bx.free()
            by.free()
        ctx.set_option("pair_small", 0)
print(json.dumps(rows, indent=1))
