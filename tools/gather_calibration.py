#!/usr/bin/env python3
"""Calibration of the fabric byte counters on GATHERS (VERDICT r5 "Next 3"): the engine's gather kernel
(k_bench_gather, the lane layout of the recurrence-step kernels) fetching UNIQUE rows of 64 / 128 / 256 / 512 bytes -
every row of a power-of-two panel exactly once per launch, scattered - so the true bytes of a launch are known exactly:
rows x row_bytes of payload + 4 bytes of index per row.  Run as is it prints the launches and their true bytes; under
`rocprofv3 --pmc <counters>` (tools/gpu_gather_calibration.sh) the per-dispatch counters of k_bench_gather<LPR, IF>
divided by these bytes are the calibration (tools/gather_calibration_summary.py).
    python tools/gather_calibration.py [log2_rows]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsp_amd import engine  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    rows = 1 << lg
    ctx = engine.default_context(0)
    out = {"rows": rows, "copy_GBps": ctx.bench_copy(1 << 29, 2), "launches": []}
    for row_bytes in (64, 128, 256, 512):
        for in_flight in (8,):
            # iters = 2 (+ 1 warm-up launch inside the call): three identical dispatches per width
            ms, gbps = ctx.bench_gather(rows, row_bytes, rows, in_flight, 0, 0.0, 8, 2)
            out["launches"].append({"row_bytes": row_bytes, "lanes_per_row": row_bytes // 16, "in_flight": in_flight,
                                    "kernel": "k_bench_gather<{}, {}>".format(row_bytes // 16, in_flight),
                                    "rows": rows, "payload_bytes": rows * row_bytes, "index_bytes": rows * 4,
                                    "panel_MB": rows * row_bytes / 1e6, "ms": ms, "payload_GBps": gbps,
                                    "rows_per_us": rows / ms / 1e3})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
