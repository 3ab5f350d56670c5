cd $GRAFT_REPO_ROOT
B='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], b["dtype"], round(b["roofline"]["avg_launch_ms"],4), round(b["roofline"]["frac"],4))'
for i in 1 2; do
  timeout 100 python bench.py --steps 8 --warmup 2 --no-cpu --no-newton --no-e2e 2>/dev/null | python -c "$B" chunked
  GSPX_VMM_CHUNK_MB=0 timeout 100 python bench.py --steps 8 --warmup 2 --no-cpu --no-newton --no-e2e 2>/dev/null | python -c "$B" hipMalloc
done
