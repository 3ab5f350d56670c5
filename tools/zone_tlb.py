#!/usr/bin/env python3
"""Do the memory zones of profiles/r06_placement.md show in the address-translation counters?  The headline call on
`candidates` backings drawn at a stride (gspx_ctx_tune_placement: three calls = 90 k_step_tile launches per candidate,
in order), to be run under `rocprofv3 --pmc <TLB counters>`; tools/zone_tlb_summary.py groups the per-dispatch rows by
candidate and sets mean launch duration beside the mean counter values.
usage: rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum -d out -o pmc --output-format csv -- \
           python tools/zone_tlb.py [candidates] [stride_mb]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsp_amd import engine, filters, graphs  # noqa: E402


def main():
    cand = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    ctx = engine.default_context(0)
    G = graphs.Sensor(1000000, k=8, seed=42)
    G.estimate_lmax("bounds")
    c = filters.compute_cheby_coeff(filters.Heat(G, 50), m=30)
    dev = G.device_graph()
    x = np.random.default_rng(0).standard_normal((G.N, 64))
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    rep = dev.tune_placement(c, bx.ptr, by.ptr, 64, float(G.lmax), cand, stride)
    print(json.dumps({"launch_ms": [round(v, 4) for v in rep["launch_ms"]], "kept": rep["kept"],
                      "candidates": cand, "stride_mb": stride}))


if __name__ == "__main__":
    main()
