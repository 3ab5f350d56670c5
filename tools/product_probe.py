#!/usr/bin/env python3
"""The product-form evaluation on the headline graph: parity against the oracle (2 columns) and launch times beside
the recurrence and the Newton form, for several kernels / orders / both dtypes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cheby_oracle as orc  # noqa: E402
from pygsp_amd import engine, filters, graphs  # noqa: E402

ctx = engine.default_context(0)
G = graphs.Sensor(1000000, k=8, seed=42)
G.estimate_lmax("bounds")
L = G.L.astype(np.float64)
x = np.random.default_rng(0).standard_normal((G.N, 64)).astype(np.float32)
banks = {"heat50": filters.Heat(G, 50), "heat10": filters.Heat(G, 10), "mh0": filters.MexicanHat(G, Nf=6)[0],
         "mh3": filters.MexicanHat(G, Nf=6)[3]}
for dt in (np.float64, np.float32):
    dev = G.device_graph(dt)
    xs = x.astype(dt)
    bx, by = ctx.upload(xs), ctx.alloc(xs.nbytes)
    for name, bank in banks.items():
        for order in (30, 100):
            c = np.atleast_2d(filters.compute_cheby_coeff(bank, m=order))
            ref = orc.cheby_op(L, G.lmax, c[0], x[:, :2].astype(np.float64))
            row = {"dtype": np.dtype(dt).name, "kernel": name, "order": order}
            for how in ("recurrence", "newton", "product"):
                try:
                    if how == "recurrence":
                        f = lambda: dev.cheby_filter_dev(c, bx.ptr, by.ptr, 64, G.lmax)
                    elif how == "newton":
                        if not filters.newton_guard(c[0], dt)[1].get("finite"):
                            raise ValueError("not representable")
                        nd = filters.cheb_to_newton(c[0])
                        f = lambda: dev.newton_filter_dev(nd[0], nd[1], bx.ptr, by.ptr, 64, G.lmax)
                    else:
                        ok, m = filters.product_guard(c[0], dt)
                        row["product_guard"] = {"ok": ok, "reason": m.get("reason"), "S": m.get("S"),
                                                "A": float("%.2g" % m.get("amplification", 0)),
                                                "passes": round(m.get("panel_passes_per_order", 0), 2)}
                        prog = filters.cheb_to_product(c[0], dt)
                        f = lambda: dev.program_filter_dev(prog, bx.ptr, by.ptr, 64, G.lmax)
                    f()
                    ms = []
                    for _ in range(4):
                        f()
                        ms.append(ctx.last_timing()["steps_ms"])
                    y = by.download(xs.shape, dt)[:, :2].astype(np.float64)
                    row[how] = {"call_ms": round(float(np.median(ms)), 3),
                                "err": float("%.2g" % (np.max(np.abs(y - ref)) / np.max(np.abs(ref))))}
                except Exception as e:
                    row[how] = {"error": repr(e)[:80]}
            print(json.dumps(row), flush=True)
    bx.free()
    by.free()
