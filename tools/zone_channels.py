#!/usr/bin/env python3
"""Per placement candidate of tools/zone_tlb.py: how evenly a counter's per-instance values (the 128 L2 channels of
TCC_* counters) are loaded - max / mean and std / mean over the instances of every wide-step launch, averaged per
candidate - from the JSON output of `rocprofv3 --pmc <raw counters> --output-format json`.
usage: tools/zone_channels.py <dir with *_results.json> [launches per candidate = 90]"""
import glob
import json
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    files = glob.glob(os.path.join(d, "**", "*_results.json"), recursive=True)
    if not files:
        print("no *_results.json under", d)
        return
    tool = json.load(open(files[0]))["rocprofiler-sdk-tool"][0]
    names = {}
    for c in tool.get("counters", []):
        h = c.get("id", {}).get("handle")
        if h is not None:
            names[h] = c.get("name")
    wide = {k["kernel_id"] for k in tool["kernel_symbols"]
            if "k_step_tile<double" in (k.get("formatted_kernel_name") or k.get("demangled_kernel_name") or k.get("kernel_name") or "")}
    # (every recurrence launch of a call, like tools/zone_tlb_summary.py: 30 per call, 90 per candidate)
    recs = tool["callback_records"].get("counter_collection") or tool["buffer_records"].get("counter_collection")
    disp = []
    for r in recs:
        dd = r["dispatch_data"]
        if dd["dispatch_info"]["kernel_id"] not in wide:
            continue
        by = {}
        for e in r["records"]:
            h = e["counter_id"]["handle"]
            # (the handle carries the instance in its upper bits on some versions: group by the counter's name if known)
            by.setdefault(names.get(h, h), []).append(e["value"])
        disp.append((dd["dispatch_info"]["dispatch_id"], dd["end_timestamp"] - dd["start_timestamp"], by))
    disp.sort()
    n = len(disp) // per
    keys = sorted({k for _, _, by in disp for k in by}, key=str)
    print("wide-step launches:", len(disp), "=", n, "candidates of", per, "; instances per counter:",
          {str(k): len(disp[0][2][k]) for k in keys})
    print("cand  launch_us  " + "  ".join("{}: sum max/mean std/mean".format(k) for k in keys))
    table = []
    for i in range(n):
        chunk = disp[i * per + per // 3:(i + 1) * per]
        row = [np.mean([c[1] for c in chunk]) / 1e3]
        for k in keys:
            v = np.array([c[2][k] for c in chunk], dtype=np.float64)  # launches x instances
            m = v.mean(axis=1)
            row += [v.sum(axis=1).mean(), float(np.mean(v.max(axis=1) / np.maximum(m, 1e-30))),
                    float(np.mean(v.std(axis=1) / np.maximum(m, 1e-30)))]
        table.append(row)
        print("{:4d}  {:9.2f}  ".format(i, row[0]) + "  ".join("{:.4g} {:.4f} {:.4f}".format(*row[1 + 3 * j:4 + 3 * j])
                                                            for j in range(len(keys))))
    t = np.array(table)
    if n >= 4:
        for j, k in enumerate(keys):
            for o, what in ((0, "sum"), (1, "max/mean"), (2, "std/mean")):
                col = t[:, 1 + 3 * j + o]
                if np.std(col) > 0:
                    print("corr(duration, {} {}) = {:+.3f}".format(k, what, float(np.corrcoef(t[:, 0], col)[0, 1])))


if __name__ == "__main__":
    main()
