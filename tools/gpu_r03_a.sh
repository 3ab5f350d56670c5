# round 3, first GPU job: smoke, GPU tests, the new microbenchmarks, one bench run
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
timeout 300 python tools/gather_ceiling.py > $O/r03_gather_ceiling.json 2> $O/gather_ceiling.err; echo "gather rc=$?"
timeout 300 python tools/hostpipe_sweep.py f64 > $O/r03_hostpipe_sweep_f64.json 2> $O/hostpipe.err; echo "hostpipe rc=$?"
timeout 700 python bench.py --steps 10 --warmup 3 > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 2 --no-configs > $O/bench_threads2.json 2> $O/bench_threads2.err; echo "bench threads rc=$?"
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/bench_f64.json"))
    print("value %.4g"%b["value"], "ms/step %.3f"%b["ms_per_step"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("achieved","frac","traffic","avg_launch_ms")})
    print("e2e", b.get("end_to_end_host_arrays"))
    for c in b.get("configs", []):
        print(c["key"], c["dtype"], "ms %.3f"%c["ms"], "frac %.3f"%c["roofline"]["frac"], "gather", c.get("roofline_gather") and {k: c["roofline_gather"][k] for k in ("achieved","peak","frac","bound_ms","step_ms")})
except Exception as e:
    print("bench failed", e)
try:
    g=json.load(open("gpurun_out/r03_gather_ceiling.json"))
    print("copy", g["copy_GBps"])
    for s in g["shapes"]:
        print(s["shape"], s["best"])
    h=json.load(open("gpurun_out/r03_hostpipe_sweep_f64.json"))
    for r in h["rows"]:
        print(r["config"], "%.2f ms"%r["ms"], r.get("identical_to_one_shot"), r["stages"])
    t=json.load(open("gpurun_out/bench_threads2.json"))
    print("threads2", t["value"], t["gather_ms"], t["gather_impl"], t["parity_vs_oracle"], t["per_device"])
except Exception as e:
    print("extras failed", e)
PY
