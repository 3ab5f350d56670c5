#!/usr/bin/env python3
"""Per-width table from the files tools/gpu_prof_widths.sh leaves (r04_narrow_<dt>_<W>_rocprofv3_summary.txt +
..._stats_bench.json): step-kernel build, its average duration over bench.py's timed launches (rocprofv3 trace and HIP
events), the width's algorithmic bytes per launch, the fraction of 8 TB/s recomputed from them, the HBM bytes per launch
from the counters (2 FETCH_SIZE + WRITE_SIZE), TCC hit rate, LDS-wait share and waves per launch.
usage: width_fracs.py <dir> <f64|f32> W [W ...]"""
import json
import os
import re
import sys

d, dt, widths = sys.argv[1], sys.argv[2], [int(w) for w in sys.argv[3:]]
elt = 8 if dt == "f64" else 4
print("| signals | row bytes | step kernel (build) | us / launch: rocprofv3 / HIP events | algorithmic MB / launch | frac of 8 TB/s "
      "| HBM MB / launch (2F + W) | traffic / algorithmic | TCC hit | LDS-wait / wave cycles | waves / launch |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for w in widths:
    base = os.path.join(d, "r04_narrow_{}_{}".format(dt, w))
    try:
        txt = open(base + "_rocprofv3_summary.txt").read()
        b = json.loads(open(base + "_stats_bench.json").read().strip().splitlines()[-1])
    except Exception as e:
        print("| {} | | missing: {} |".format(w, e))
        continue
    r = b["roofline"]
    m = re.search(r"rocprofv3 avg ([\d.]+) us, bench.py HIP events of the same launches ([\d.]+) us", txt)
    prof_us, ev_us = (float(m.group(1)), float(m.group(2))) if m else (float("nan"), r["avg_launch_ms"] * 1e3)
    per = {}
    for line in txt.splitlines():
        mm = re.match(r"^\s{3}(.{60})\s(\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        if mm and "k_step" in mm.group(1):
            per.setdefault(mm.group(1).strip(), {})[mm.group(2)] = (int(mm.group(3)), float(mm.group(4)))
    # the build that ran most launches (the step-2 flavour of a fused-input call runs once per call)
    kern = max(per, key=lambda k: max(v[0] for v in per[k].values())) if per else None
    cnt = {c: v[1] for c, v in per.get(kern, {}).items()}
    t = re.search(r'"hbm_bytes_per_launch": ([\d.e+]+)', txt)
    traffic = float(t.group(1)) if t else float("nan")
    alg = r["algorithmic_bytes_per_launch"]
    hit = cnt.get("TCC_HIT_sum", float("nan")) / max(cnt.get("TCC_HIT_sum", 0) + cnt.get("TCC_MISS_sum", 0), 1)
    ldsw = cnt.get("SQ_WAIT_INST_LDS", float("nan")) / max(cnt.get("SQ_WAVE_CYCLES", 0), 1)
    name = re.sub(r"gspx::TileArgs<\w+>", "", kern or "?").replace("void gspx::", "")[:58]
    print("| {} | {} | `{}` | {:.1f} / {:.1f} | {:.1f} | **{:.3f}** | {:.1f} | {:.2f} | {:.2f} | {:.3f} | {:.0f} |".format(
        w, w * elt, name, prof_us, ev_us, alg / 1e6, alg / (prof_us * 1e-6) / 8e12 if prof_us == prof_us else r["frac"],
        traffic / 1e6, traffic / alg, hit, ldsw, cnt.get("SQ_WAVES", float("nan"))))
