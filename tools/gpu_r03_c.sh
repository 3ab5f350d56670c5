# round 3, third GPU job: full GPU suite, host pipeline schedules, config-3 occupancy sweep, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log
timeout 300 python tools/hostpipe_sweep.py f64 auto > $O/r03_hostpipe_sweep_f64.json 2> $O/hostpipe.err; echo "hostpipe rc=$?"
timeout 300 python tools/hostpipe_sweep.py f32 auto > $O/r03_hostpipe_sweep_f32.json 2> $O/hostpipe32.err; echo "hostpipe32 rc=$?"
timeout 400 python tools/c3_occupancy.py > $O/r03_c3_occupancy.json 2> $O/c3occ.err; echo "c3occ rc=$?"
timeout 700 python bench.py --steps 10 --warmup 3 --no-configs > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/bench_f64.json"))
    print("value %.4g"%b["value"], "ms/step %.3f"%b["ms_per_step"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("achieved","frac","traffic","avg_launch_ms")})
    print("setup", b.get("setup_s"))
    print("e2e", {k:v for k,v in b.get("end_to_end_host_arrays",{}).items() if k!="note"})
except Exception as e:
    print("bench failed", e)
for f in ("gpurun_out/r03_hostpipe_sweep_f64.json","gpurun_out/r03_hostpipe_sweep_f32.json"):
    try:
        h=json.load(open(f))
        for r in h["rows"]:
            st=r["stages"] or {}
            print(h["dtype"], r["config"], "%.2f ms"%r["ms"], r.get("identical_to_one_shot"), {k:(round(v,2) if isinstance(v,float) else v) for k,v in st.items()})
    except Exception as e:
        print("extras failed", e)
try:
    rows=json.load(open("gpurun_out/r03_c3_occupancy.json"))
    for dt in ("float64","float32"):
        print(dt)
        for pad in (0,8,16,29,40):
            print("  pad",pad," ".join("rpw%d:%.3f"%(r["rows_per_wave"],r["step_ms"]) for r in rows if r["dtype"]==dt and r["lds_pad_kb"]==pad))
except Exception as e:
    print("c3occ failed", e)
PY
