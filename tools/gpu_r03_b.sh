# round 3, second GPU job: tests, pipeline sweep, gather ceiling with counters, bench, rocprofv3 evidence
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log
timeout 300 python tools/hostpipe_sweep.py f64 auto > $O/r03_hostpipe_sweep_f64.json 2> $O/hostpipe.err; echo "hostpipe rc=$?"
timeout 700 python bench.py --steps 10 --warmup 3 > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench rc=$?"
export TMPDIR=/tmp
( cd /tmp
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $O/gather_pmc_$name -o pmc -- python $R/tools/gather_ceiling.py --quick > /dev/null 2> $O/gather_pmc_$name.err
done )
python - <<'PY' > $O/r03_gather_ceiling_counters.txt 2>&1
import csv, glob, os
from collections import defaultdict
for d in sorted(glob.glob("gpurun_out/gather_pmc_*")):
    if not os.path.isdir(d): continue
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_bench_gather<" in r["Kernel_Name"] or "k_permute_in" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    print("--", d)
    for (k, c, g), v in sorted(acc.items()):
        # launches come in the order of tools/gather_ceiling.py's SHAPES: warm-up + 3 timed each
        print("  {:60s} {:20s} n={:3d} per-launch: {}".format(k, c, len(v), " ".join("%.4g" % x for x in v[:24])))
PY
bash tools/gpu_prof.sh f64 --dtype f64 > $O/prof_f64.log 2>&1; tail -40 $O/prof_f64.log
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/bench_f64.json"))
    print("value %.4g"%b["value"], "ms/step %.3f"%b["ms_per_step"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("achieved","frac","traffic","avg_launch_ms")})
    print("setup", b.get("setup_s"))
    print("e2e", {k:v for k,v in b.get("end_to_end_host_arrays",{}).items() if k!="note"})
    for c in b.get("configs", []):
        print(c["key"], c["dtype"], "ms %.3f"%c["ms"], "frac %.3f"%c["roofline"]["frac"], "gather", c.get("roofline_gather") and {k: c["roofline_gather"][k] for k in ("achieved","peak","frac","bound_ms","step_ms")})
except Exception as e:
    print("bench failed", e)
try:
    h=json.load(open("gpurun_out/r03_hostpipe_sweep_f64.json"))
    for r in h["rows"]:
        print(r["config"], "%.2f ms"%r["ms"], r.get("identical_to_one_shot"), r["stages"])
except Exception as e:
    print("extras failed", e)
PY
