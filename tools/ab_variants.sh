# one box, one build: k_step_tile software-pipelining variants (option tile_variant) against the previous build
q() { timeout 300 python bench.py --no-cpu --no-configs --no-newton --no-e2e --no-live-traffic --steps 10 --warmup 3 "$@" | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print(round(r['frac'],4), round(r['avg_launch_ms'],4))"; }
for shape in "--dtype f32" "--dtype f64 --nsig 32 --vertices 500000" "--dtype f32 --nsig 128 --vertices 500000" "--dtype f64"; do
  for round in 1 2; do
    echo "$shape old: $(GSPX_LIB_PATH=pygsp_amd/_lib/libgspx_old.so q $shape)"
    for v in 0 1 5 13; do echo "$shape variant $v: $(q $shape --opt tile_variant=$v)"; done
  done
done
