R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python -m pytest tests -m gpu -q -x --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
SECONDS=0
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
b=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.4g"%b["value"], json.dumps(b["roofline"], indent=None)[:1500])
print("config", {k:b["config"][k] for k in ("gather_impl","rccl_version","rccl_nranks_seen","launcher_world_size")})
print("cpu", b["cpu_baseline"]["kind"], b["cpu_baseline"]["value"])
PY
timeout 900 bash tools/gpu_real_pygsp.sh > $O/real_pygsp_stdout.log 2>&1; echo "real pygsp rc=$?"; tail -40 $O/real_pygsp_gpu.log
