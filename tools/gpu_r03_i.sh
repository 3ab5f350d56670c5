# round 3: full GPU suite
# (the host-sanitizer build cannot run here: the HIP runtime aborts at device initialisation under ASan on this
# stack, so tests/test_asan.py covers the host paths that need no device)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 2 > $O/bench_threads2.json 2> $O/bench_threads2.err; echo "bench threads rc=$?"
python - <<'PY'
import json
t=json.load(open("gpurun_out/bench_threads2.json"))
print("threads2 value %.4g"%t["value"], "gather_ms", t["gather_ms"], t["gather_impl"][:60], t["parity_vs_oracle"])
print("signal_parallel", {k:(round(v,3) if isinstance(v,float) else v) for k,v in t["signal_parallel"].items() if k!="workload"}, t["signal_parallel"].get("workload"))
print("batch4", {k:(round(v,3) if isinstance(v,float) else v) for k,v in t["batch_config4"].items() if k not in ("workload",)})
PY
