# A/B of two builds of libgspx on ONE box (boxes differ by several percent): alternate, several rounds.
# usage: bash tools/ab_bench.sh <old.so> [bench args]
OLD=$1; shift
q() { timeout 300 python bench.py --no-cpu --no-configs --no-newton --no-mix --no-e2e --no-f32 --no-live-traffic --steps 10 --warmup 3 "$@" | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print(round(r['frac'],4), round(r['avg_launch_ms'],4))"; }
for round in 1 2 3; do
  echo "old $* : $(GSPX_LIB_PATH=$OLD q "$@")"
  echo "new $* : $(q "$@")"
done
