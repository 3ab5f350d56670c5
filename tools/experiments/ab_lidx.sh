cd $GRAFT_REPO_ROOT
cp pygsp_amd/_lib/libgspx.so /tmp/new.so; cp pygsp_amd/_lib/libgspx_old.so /tmp/old.so
for rep in 1 2; do for v in old new; do cp /tmp/$v.so pygsp_amd/_lib/libgspx.so
for d in f64 f32; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-newton --no-e2e --dtype $d 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', b['dtype'], round(b['value']/1e9,1), round(b['roofline']['avg_launch_ms'],4))"; done; done; done
