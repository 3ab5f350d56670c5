# round 3, seventh GPU job: k-NN (new sweep kernel), host pipeline with pre-faulting, default bench run timed
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_5_knn.py tests/test_gpu_2_kernels.py tests/test_gpu_9_fuzz.py -m gpu -q > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_part.log
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_knn_hd -o stats -- python $R/tools/knn_highdim_bench.py > $O/r03_knn_highdim_bench.json 2> $O/knn_bench.err ); echo "knn prof rc=$?"
timeout 300 python tools/hostpipe_sweep.py f64 auto > $O/r03_hostpipe_sweep_f64.json 2> $O/hostpipe.err; echo "hostpipe rc=$?"
SECONDS=0
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"
python - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/prof_knn_hd/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("{:80.80s} calls={:>5} total_ms={:>10.3f} avg_us={:>10.1f} pct={}".format(r["Name"], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
for c in json.load(open("gpurun_out/r03_knn_highdim_bench.json"))["cases"]:
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in c.items() if k in ("N","d","k","device_build_ms","pair_distances_per_s","mfma_flops_per_s","mean_candidates","exact_scans","neighbours_identical","distances_identical")})
h=json.load(open("gpurun_out/r03_hostpipe_sweep_f64.json"))
for r in h["rows"]:
    st=r["stages"] or {}
    print(r["config"], "%.2f ms"%r["ms"], r.get("identical_to_one_shot"), {k:(round(v,2) if isinstance(v,float) else v) for k,v in st.items()})
b=json.load(open("gpurun_out/bench_default.json"))
print("value %.4g"%b["value"], "frac %.4f"%b["roofline"]["frac"], "traffic", b["roofline"]["traffic"])
e=b["end_to_end_host_arrays"]; print("e2e %.2f one-shot %.1f"%(e["ms"], e["one_shot_ms"]), e["stages"])
print("setup", b["setup_s"])
for c in b.get("configs", []):
    print(c["key"], c["dtype"], "ms %.3f"%c["ms"], "frac %.3f"%c["roofline"]["frac"], "gather", c.get("roofline_gather") and round(c["roofline_gather"]["frac"],3))
print("batch4", {k:b["batch_config4"][k] for k in ("ms","value","gather_ms")})
PY
