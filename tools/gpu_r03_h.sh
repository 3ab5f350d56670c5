R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_5_knn.py -m gpu -q > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_part.log
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_knn_hd -o stats -- python $R/tools/knn_highdim_bench.py > $O/r03_knn_highdim_bench.json 2> $O/knn_bench.err ); echo "knn prof rc=$?"
python - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/prof_knn_hd/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("{:80.80s} calls={:>5} total_ms={:>10.3f} avg_us={:>10.1f} pct={}".format(r["Name"], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
for c in json.load(open("gpurun_out/r03_knn_highdim_bench.json"))["cases"]:
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in c.items() if k in ("N","d","k","device_build_ms","pair_distances_per_s","mfma_flops_per_s","capacity","mean_candidates","exact_scans","neighbours_identical","distances_identical","host_s_extrapolated_to_N")})
PY
