# round 3, sixth GPU job: config-1 latency sweep, end-to-end diagnosis, timing of the default bench run
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/c1_latency.py > $O/r03_c1_latency.json 2> $O/c1.err; echo "c1 rc=$?"
timeout 400 python bench.py --steps 5 --warmup 2 --no-configs --no-newton --cpu-all-cores 0 --no-live-traffic > $O/bench_e2e_nopool.json 2> $O/bench_e2e_nopool.err; echo "bench nopool rc=$?"
timeout 400 python bench.py --steps 5 --warmup 2 --no-configs --no-newton --no-live-traffic > $O/bench_e2e_pool.json 2> $O/bench_e2e_pool.err; echo "bench pool rc=$?"
/usr/bin/time -v -o $O/bench_default.time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
grep -E "Elapsed|Maximum resident" $O/bench_default.time
python - <<'PY'
import json
rows=json.load(open("gpurun_out/r03_c1_latency.json"))
for g in (-1,0,1,2,3,4,5):
    print("glog2",g," ".join("rpw%d:%.4f"%(r["rows_per_wave"],r["best_ms"]) for r in rows if r["narrow_g_log2"]==g))
for f in ("bench_e2e_nopool","bench_e2e_pool","bench_default"):
    try:
        b=json.load(open("gpurun_out/%s.json"%f))
        e=b["end_to_end_host_arrays"]
        print(f, "frac %.4f"%b["roofline"]["frac"], "e2e %.2f ms"%e["ms"], "one-shot %.1f"%e["one_shot_ms"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in (e["stages"] or {}).items()})
    except Exception as ex:
        print(f, "failed", ex)
PY
