# end-of-iteration GPU job: smoke, tests, rocprofv3 evidence (headline both dtypes, configs 2/3), bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $R/gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $R/gpurun_out/pytest_gpu.log
bash tools/gpu_prof.sh f64 --dtype f64 > /dev/null 2>&1
bash tools/gpu_prof.sh f32 --dtype f32 > /dev/null 2>&1
cp $R/gpurun_out/prof_f64/traffic.json $R/profiles/traffic_f64.json 2>/dev/null
cp $R/gpurun_out/prof_f32/traffic.json $R/profiles/traffic_f32.json 2>/dev/null
bash tools/gpu_prof_configs.sh c2 > /dev/null 2>&1
bash tools/gpu_prof_configs.sh c3 > /dev/null 2>&1
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --dtype f64 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err; echo "bench f64 rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --dtype f32 > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; echo "bench f32 rc=$?"
cp profiles/traffic_f64.json profiles/traffic_f32.json gpurun_out/ 2>/dev/null
python - <<'PY'
import json
for t in ("f64","f32"):
    try:
        b=json.load(open("gpurun_out/bench_%s.json"%t))
        print(t, "value %.4g"%b["value"], "ms/step %.3f"%b["ms_per_step"], "roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("achieved","frac","traffic","avg_launch_ms")}, "cpu", b.get("cpu_baseline",{}).get("value"), "err", b.get("parity_vs_oracle",{}).get("max_rel_err"))
    except Exception as e:
        print(t, "failed", e)
PY
