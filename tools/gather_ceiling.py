#!/usr/bin/env python3
"""Random-row-gather ceiling of this box (gspx_bench_gather) for the shapes of BASELINE configs 2 and 3 and the
headline row width: every in-flight depth x workgroups per CU, uniform and block-local indices.
    python tools/gather_ceiling.py > gpurun_out/r03_gather_ceiling.json
Under rocprofv3 --pmc (one counter set per pass) the same command gives FETCH_SIZE / TCC_HIT / TCC_MISS per launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine  # noqa: E402

SHAPES = [  # name, panel rows, row bytes, gathers per step, blocks, p_intra
    ("c2 ER 1e6 x 64 fp32", 1000000, 256, 11000000, 1, 0.0),
    ("c3 SBM 2e6 x 16 fp64 (uniform)", 2000000, 128, 34000000, 1, 0.0),
    ("c3 SBM 2e6 x 16 fp64 (block-local 0.75)", 2000000, 128, 34000000, 16, 0.75),
    ("c3 SBM 2e6 x 16 fp32 (uniform)", 2000000, 64, 34000000, 1, 0.0),
    ("c3 SBM 2e6 x 16 fp32 (block-local 0.75)", 2000000, 64, 34000000, 16, 0.75),
    ("headline width 1e6 x 64 fp64 (uniform)", 1000000, 512, 10200000, 1, 0.0),
]


def main():
    quick = "--quick" in sys.argv
    ctx = engine.default_context(0)
    out = {"copy_GBps": ctx.bench_copy(1 << 30, 5), "shapes": []}
    for name, rows, rb, n, blocks, p in SHAPES:
        grid = []
        for wg in ((8,) if quick else (2, 4, 8, 16)):
            for depth in ((8,) if quick else (2, 4, 8, 16)):
                ms, gbps = ctx.bench_gather(rows, rb, n, depth, blocks, p, wg, 3)
                grid.append({"workgroups_per_cu": wg, "in_flight": depth, "ms": ms, "GBps": gbps,
                             "Grows_per_s": n / ms / 1e6})
        best = max(grid, key=lambda r: r["GBps"])
        out["shapes"].append({"shape": name, "panel_MB": rows * rb / 1e6, "row_bytes": rb, "gathers": n,
                              "best": best, "grid": grid})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
