#!/usr/bin/env python3
"""The headline graph (Sensor 1M, k = 8, Heat, order 30) against the panel width: what a GPU of a signal-parallel
group (64 columns over 8 GPUs = 8 per GPU) and a batch of the host pipeline see.  Per width: device time of the call,
the width's own algorithmic bytes K (CSR + 3U) + U and the fraction of the 8 TB/s peak.  GPU box only."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    K = 30
    ctx = engine.default_context(0)
    for kv in sys.argv[2:]:  # context options, e.g. tile_lg=8 (the 8-lane build for every narrow panel)
        key, val = kv.split("=")
        ctx.set_option(key, int(val))
    rows = []
    for dtype in (np.float64, np.float32):
        G = graphs.Sensor(n, k=8, seed=42, compute_dtype=dtype)
        G.estimate_lmax("bounds")
        c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
        dev = G.device_graph()
        elt = np.dtype(dtype).itemsize
        x = np.random.default_rng(0).standard_normal((G.N, 64)).astype(dtype)
        widths = [int(v) for v in os.environ.get("WIDTHS", "1,2,3,4,5,6,8,10,12,16,24,32,48,64").split(",")]
        ab = os.environ.get("AB")  # an option measured at 0 and at 1 for every width, alternating (e.g. AB=tile_regroup)
        for w in widths:
            xs = np.ascontiguousarray(x[:, :w])
            bx, by = ctx.upload(xs), ctx.alloc(xs.nbytes)
            y_first = None
            for setting in ((0, 1, 0, 1) if ab else (None,)):
                if ab:
                    ctx.set_option(ab, setting)
                ms = [dev.cheby_filter_dev(c, bx.ptr, by.ptr, w, float(G.lmax)) for _ in range(12)]
                t = ctx.last_timing()
                y = by.download(xs.shape, dtype)  # both settings must produce the same result
                if y_first is None:
                    y_first = y
                diff = float(np.max(np.abs(y - y_first)) / max(float(np.max(np.abs(y_first))), 1e-300))
                best = float(np.median(ms[3:]))
                U = G.N * w * elt
                b_alg = K * (dev.nnz_l * (elt + 4) + 4 * (G.N + 1) + 3 * U) + U
                rows.append({"dtype": np.dtype(dtype).name, "signals": w, "row_bytes": w * elt, "ms": best,
                             "ms_per_signal": best / w, "alg_GB": b_alg / 1e9, "frac_8TBs": b_alg / (best * 1e-3) / 8e12,
                             "steps_ms": t["steps_ms"], "step_launches": t["step_launches"]})
                if ab:
                    rows[-1][ab] = setting
                    rows[-1]["rel_diff_vs_setting_0"] = diff
            bx.free()
            by.free()
        G.__dict__.pop("_dev", None)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
