# the evidence part of tools/gpu_final.sh without the test run (for the last minutes of a GPU budget)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
bash tools/gpu_prof.sh f64 --dtype f64 > /dev/null 2>&1
bash tools/gpu_prof.sh f32 --dtype f32 > /dev/null 2>&1
cd $R
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f64 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err; echo "bench f64 rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; echo "bench f32 rc=$?"
python - <<'PY'
import json
for t in ("f64","f32"):
    b=json.load(open("gpurun_out/bench_%s.json"%t))
    print(t, "value %.4g"%b["value"], "frac %.4f"%b["roofline"]["frac"], "avg %.4f"%b["roofline"]["avg_launch_ms"], "traffic", b["roofline"]["traffic"], "err", b.get("parity_vs_oracle",{}).get("max_rel_err"))
PY
