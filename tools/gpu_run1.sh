mkdir -p gpurun_out
ls /root/reference 2>&1 | head -2 > gpurun_out/probe.log
rocm-smi --showmeminfo vram 2>&1 | tail -5 >> gpurun_out/probe.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err; echo "bench rc=$?"
cat gpurun_out/bench_f64.json | head -c 3000
timeout 900 python tools/sweep.py > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
tail -5 gpurun_out/sweep.log
