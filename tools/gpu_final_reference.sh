R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
PYGSP_PATH=$R/_ref_stage timeout 600 python -m pytest tests/test_gpu_b_real_pygsp.py -m gpu -q -rs > $O/pytest_real_pygsp.log 2>&1; echo "real pygsp rc=$?"; tail -3 $O/pytest_real_pygsp.log
SECONDS=0
PYGSP_PATH=$R/_ref_stage timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'PY'
import json
b=json.loads(open("gpurun_out/bench_driver_form.json").read().strip().splitlines()[-1])
r=b["roofline"]
print("value %.4g ms_per_step %.3f"%(b["value"], b["ms_per_step"]), {k:r[k] for k in ("frac","frac_whole_call","traffic_over_algorithmic","f32_frac","newton_frac","parity_max_rel_err","configs_frac")})
print("cpu", b["cpu_baseline"]["kind"], b["cpu_baseline"]["value"], b.get("parity_vs_reference"))
print("config", {k:b["config"][k] for k in ("gather_impl","rccl_version","rccl_nranks_seen","launcher_world_size")})
for c in b.get("configs", []):
    print(c["key"], c["dtype"], "frac %.3f"%c["roofline"]["frac"], "traffic/alg", c["roofline"].get("traffic_over_algorithmic"))
PY
