# round 3, fourth GPU job: full GPU suite (+ timing of the high-dimensional k-NN search)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest_gpu.log
timeout 900 python tools/knn_highdim_bench.py > $O/r03_knn_highdim_bench.json 2> $O/knn_bench.err; echo "knn bench rc=$?"; python -c "import json; [print({k:(round(v,3) if isinstance(v,float) else v) for k,v in c.items()}) for c in json.load(open(\"gpurun_out/r03_knn_highdim_bench.json\"))[\"cases\"]]"
