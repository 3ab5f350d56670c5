#!/bin/bash
# The driver's own command on a fresh box (python bench.py, no flags) and its one-line summary:
# gpurun -- 'bash tools/gpu_bench_default.sh'; the lines are collected in profiles/r06_bench_default_boxes.jsonl
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench_default_run.json 2> gpurun_out/bench_default_run.err
echo "rc=$? wall=$(( $(date +%s) - t0 ))s"
python tools/bench_default_line.py gpurun_out/bench_default_run.json "$(date -u +%H:%M)" | tee gpurun_out/bench_default_line.json
