#!/usr/bin/env python3
"""Per placement candidate of tools/zone_tlb.py: mean k_step_tile duration and mean counter values from a rocprofv3
--pmc counter_collection.csv (rows in dispatch order, 90 wide-step launches per candidate), and their correlation.
usage: tools/zone_tlb_summary.py <dir with *counter_collection.csv> [launches per candidate = 90]"""
import csv
import glob
import os
import sys
from collections import OrderedDict

import numpy as np


def main():
    d = sys.argv[1]
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    rows = OrderedDict()  # dispatch id -> {"dur": ns, counter: value}
    for r in csv.DictReader(open(files[0])):
        if "k_step_tile<double" not in r["Kernel_Name"]:
            continue
        e = rows.setdefault(int(r["Dispatch_Id"]), {})
        e["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = [rows[k] for k in sorted(rows)]
    names = sorted(k for k in disp[0] if k != "dur")
    n = len(disp) // per
    print("wide-step launches:", len(disp), "=", n, "candidates of", per)
    print("cand  launch_us  " + "  ".join(names))
    table = []
    for i in range(n):
        chunk = disp[i * per + per // 3:(i + 1) * per]  # the first call of a candidate allocates: skip it
        dur = np.mean([e["dur"] for e in chunk]) / 1e3
        vals = [np.mean([e.get(k, 0.0) for e in chunk]) for k in names]
        table.append([dur] + vals)
        print("{:4d}  {:9.2f}  ".format(i, dur) + "  ".join("{:.4g}".format(v) for v in vals))
    t = np.array(table)
    if n >= 4:
        for j, k in enumerate(names):
            if np.std(t[:, j + 1]) > 0:
                print("corr(duration, {}) = {:+.3f}   (spread of the counter: {:.3g} ... {:.3g})".format(
                    k, float(np.corrcoef(t[:, 0], t[:, j + 1])[0, 1]), t[:, j + 1].min(), t[:, j + 1].max()))
            else:
                print("{}: constant {:.4g}".format(k, t[0, j + 1]))


if __name__ == "__main__":
    main()
