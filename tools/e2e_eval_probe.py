import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygsp_amd import filters, graphs
G = graphs.Sensor(1000000, k=8, seed=42); G.estimate_lmax("bounds")
x = np.random.default_rng(0).standard_normal((G.N, 64))
filters.AUTO_MAX_HOST_PANEL_BYTES = 1 << 60
for name, bank, order in (("heat50", filters.Heat(G, 50), 30), ("heat10", filters.Heat(G, 10), 100), ("heat50", filters.Heat(G, 50), 100)):
    row = {"kernel": name, "order": order}
    ys = {}
    for ev in ("recurrence", "auto", "newton"):
        bank.filter(x, order=order, evaluation=ev)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); y = bank.filter(x, order=order, evaluation=ev); best = min(best, time.perf_counter() - t)
        ys[ev] = y
        row[ev] = {"ms": round(best * 1e3, 1), "how": G._gspx_last_evaluation}
    row["auto_vs_recurrence_rel_diff"] = float(np.max(np.abs(ys["auto"] - ys["recurrence"])) / np.max(np.abs(ys["recurrence"])))
    print(json.dumps(row), flush=True)
