q() { timeout 300 python bench.py --no-cpu --no-configs --no-newton --no-mix --no-e2e --no-f32 --steps 10 --warmup 3 | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print('$1', round(r['frac'],4), round(r['avg_launch_ms'],4), 'copy', round(r['copy_GBps_this_run'],0))"; }
mkdir -p gpurun_out
{
q fresh1; q fresh2
timeout 600 python -m pytest tests/test_gpu_8_alloc.py tests/test_gpu_1_configs.py -q -m gpu 2>&1 | tail -1
q after_tests1; q after_tests2
sleep 45
q after_sleep
rocm-smi --showtemp --showpower 2>&1 | grep -E "Temperature|Power \(W\)"
} 2>&1 | tee gpurun_out/bench_variance2.log
