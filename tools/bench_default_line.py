#!/usr/bin/env python3
"""One line of what the driver's default `python bench.py` reported on this box: card, headline fraction (tuned / first
draw / kept candidate), mix ceiling, evaluation='auto', fp32.  usage: tools/bench_default_line.py bench.json [label]"""
import json, sys

b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = b["roofline"], b["config"]
pt = (b.get("setup_s") or {}).get("placement_tuning") or {}
smi = r.get("smi_under_load") or c.get("smi") or {}
print(json.dumps({
    "label": sys.argv[2] if len(sys.argv) > 2 else "",
    "card": smi.get("gpu_unique_id"),
    "frac": round(r["frac"], 4), "frac_whole_call": round(r.get("frac_whole_call") or 0, 4),
    "first_draw": round(r.get("frac_untuned_first_draw") or 0, 4), "kept": pt.get("kept"),
    "fast_candidates": sum(1 for f in (r.get("placement_candidates_frac") or []) if f and f >= 0.60),
    "candidates": sum(1 for f in (r.get("placement_candidates_frac") or []) if f),
    "frac_of_mix_ceiling": round(r.get("frac_of_mix_ceiling") or 0, 3),
    "auto": r.get("auto_evaluation"), "auto_frac": round(r.get("auto_frac") or 0, 4),
    "auto_parity": r.get("auto_parity_max_rel_err"),
    "f32_frac": round(r.get("f32_frac") or 0, 4), "f32_auto_frac": round(r.get("f32_auto_frac") or 0, 4),
    "parity": (b.get("parity_vs_oracle") or {}).get("max_rel_err"),
    "tuning_s": round(pt.get("seconds") or 0, 1)}))
