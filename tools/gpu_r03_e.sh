# round 3, fifth GPU job: full GPU suite, kernel split of the high-dimensional k-NN search
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_gpu.log
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_knn_hd -o stats -- python $R/tools/knn_highdim_bench.py > $O/r03_knn_highdim_bench.json 2> $O/knn_bench.err ); echo "knn prof rc=$?"
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_knn_hd/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("{:80.80s} calls={:>5} total_ms={:>10.3f} avg_us={:>10.1f} pct={}".format(r["Name"], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
import json
for c in json.load(open("gpurun_out/r03_knn_highdim_bench.json"))["cases"]:
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in c.items()})
PY
