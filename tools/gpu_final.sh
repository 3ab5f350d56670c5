# end-of-round GPU job: smoke, the whole GPU suite, the default bench line (timed), rocprofv3 evidence of the
# headline kernel (kernel trace + separate PMC passes), the threaded two-context bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1700 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
SECONDS=0
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"
bash tools/gpu_prof.sh f64 --dtype f64 > $O/prof_f64.log 2>&1; grep -E "step-kernel launches|k_step_tile<double, 2, 16, false, false.*calls" $O/prof_f64.log | head -3
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 2 > $O/bench_threads2.json 2> $O/bench_threads2.err; echo "bench threads rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/bench_default.json"))
print("value %.4g"%b["value"], "frac %.4f"%b["roofline"]["frac"], "traffic", b["roofline"]["traffic"], "avg_launch_ms", b["roofline"]["avg_launch_ms"])
e=b["end_to_end_host_arrays"]; print("e2e %.2f one-shot %.1f"%(e["ms"], e["one_shot_ms"]), e["stages"])
print("setup", b["setup_s"])
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["multi_core"]["value"], "parity", b["parity_vs_oracle"])
for c in b.get("configs", []):
    print(c["key"], c["dtype"], "ms %.3f"%c["ms"], "frac %.3f"%c["roofline"]["frac"], "gather", c.get("roofline_gather") and round(c["roofline_gather"]["frac"],3), "err", c["parity_vs_oracle"]["max_rel_err"])
print("batch4", {k:b["batch_config4"][k] for k in ("ms","value")})
t=json.load(open("gpurun_out/bench_threads2.json"))
print("threads2 value %.4g"%t["value"], t["parity_vs_oracle"], "signal_parallel %.4g"%t["signal_parallel"]["value"], "batch4 %.4g"%t["batch_config4"]["value"])
PY
