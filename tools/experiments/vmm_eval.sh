cd $GRAFT_REPO_ROOT
B='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], b["dtype"], round(b["value"]/1e9,1), round(b["roofline"]["avg_launch_ms"],4), round(b["roofline"]["frac"],4), b.get("parity_vs_oracle"))'
for d in f64 f32; do
  GSPX_VMM_CHUNK_MB=2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-newton --no-e2e --dtype $d 2>/dev/null | python -c "$B" vmm2
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-newton --no-e2e --dtype $d 2>/dev/null | python -c "$B" plain
done
GSPX_VMM_CHUNK_MB=2 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
GSPX_VMM_CHUNK_MB=2 timeout 200 python tools/configs_bench.py > gpurun_out/configs_vmm.log 2>&1; python - <<'PY'
import json
for c in json.load(open("gpurun_out/configs.json")):
    print(c["config"][:60], c["dtype"], "%.3f ms"%c["total_ms"], "%.1f%%"%(100*c["frac_8TBps"]))
PY
