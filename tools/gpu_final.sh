# end-of-round GPU job: smoke, the whole GPU suite, the default bench line (timed), the box probe (mix ceiling, stream
# mixes, card identity), rocprofv3 evidence of the headline kernel (kernel trace + separate PMC passes) and of configs
# 2 / 3 (refreshes profiles/traffic_configs.json's stamp), the threaded two-context bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showuniqueid --showmemvendor --showvbios | grep "^GPU" > $O/final_card.txt; cat $O/final_card.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1700 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
SECONDS=0
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"
timeout 300 python tools/box_probe.py > $O/box_probe_final.json 2> $O/box_probe_final.err; echo "probe rc=$?"
bash tools/gpu_prof.sh f64 --dtype f64 > $O/prof_f64.log 2>&1; grep -E "step-kernel launches|k_step_tile<double, 2, 16, false, false.*calls" $O/prof_f64.log | head -3
bash tools/gpu_prof_configs.sh c3 > $O/prof_c3.log 2>&1; bash tools/gpu_prof_configs.sh c2 > $O/prof_c2.log 2>&1
python tools/config_traffic.py $O/prof_c2 $O/prof_c3 > $O/config_traffic.log 2>&1; cp profiles/traffic_configs.json $O/traffic_configs.json
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 2 > $O/bench_threads2.json 2> $O/bench_threads2.err; echo "bench threads rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/bench_default.json"))
rf=b["roofline"]
print("value %.4g"%b["value"], "frac %.4f"%rf["frac"], "whole %.4f"%rf["frac_whole_call"], "traffic/alg", rf["traffic_over_algorithmic"], "avg_launch_ms", rf["avg_launch_ms"])
print("mix: of_ceiling %.3f ceiling_frac %.4f read %.0f streams %s"%(rf["frac_of_mix_ceiling"], rf["mix_ceiling_frac"], rf["read_GBps_this_run"], rf["stream_mix_GBps"]))
print("auto", rf["auto_evaluation"], "%.4f"%rf["auto_frac"], "newton parity", rf["newton_parity_max_rel_err"], "f32", rf["f32_frac"], rf["f32_auto_frac"], rf["f32_newton_parity_max_rel_err"])
print("tuning", b["setup_s"]["placement_tuning"]["candidates_frac"], "kept", b["setup_s"]["placement_tuning"]["kept"])
print("smi", rf["smi_under_load"])
e=b["end_to_end_host_arrays"]; print("e2e %.2f one-shot %.1f"%(e["ms"], e["one_shot_ms"]))
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["multi_core"]["value"], "parity", b["parity_vs_oracle"])
for c in b.get("configs", []):
    print(c["key"], c["dtype"], "ms %.3f"%c["ms"], "frac %.3f"%c["roofline"]["frac"], "stale", c["roofline"].get("traffic_recorded_stale"), "err", c["parity_vs_oracle"]["max_rel_err"])
t=json.load(open("gpurun_out/bench_threads2.json"))
print("threads2 value %.4g"%t["value"], t["parity_vs_oracle"], "keys", sorted(t["roofline"])[:8])
PY
