#!/bin/bash
# GPU box: the REAL pygsp (staged by tools/stage_reference.sh, or installed / $PYGSP_PATH) through plugin.install() on a
# real MI355X - the reference's own test_filters.py, the doctest chain device-resident - and bench.py with the
# reference itself as the CPU baseline of the same run.  Log -> gpurun_out/real_pygsp_gpu.log (copy to profiles/).
cd "$(dirname "$0")/.."
export PYGSP_PATH="${PYGSP_PATH:-$PWD/_ref_stage}"
mkdir -p gpurun_out
{
  echo "== real pygsp on the device: $(date -u +%FT%TZ)  PYGSP_PATH=$PYGSP_PATH"
  python -c "import sys; sys.path.insert(0, '$PYGSP_PATH'); import pygsp; print('pygsp', pygsp.__version__, pygsp.__file__)"
  rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3
  python -m pytest tests/test_gpu_b_real_pygsp.py -m gpu -v -rs -p no:cacheprovider 2>&1 | tail -15
  echo "== seam call counts of that run (GSPX_SEAM_REPORT)"
  python - <<'PY'
import json, os, subprocess, sys, tempfile
root = os.getcwd()
ref = os.environ["PYGSP_PATH"]
tmp = tempfile.mkdtemp()
env = dict(os.environ, PYTHONPATH=os.pathsep.join([ref, root, os.path.join(root, "tests")]), PYTHONDONTWRITEBYTECODE="1",
           GSPX_SEAM_REPORT=os.path.join(tmp, "seam.json"), MPLBACKEND="Agg")
res = subprocess.run([sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-p", "seam_device_plugin", "-q", "--no-header",
                      "-o", "addopts=", "--rootdir", tmp, os.path.join(ref, "pygsp", "tests", "test_filters.py")],
                     cwd=tmp, env=env, capture_output=True, text=True)
print(res.stdout.strip().splitlines()[-1])
print(open(os.path.join(tmp, "seam.json")).read())
PY
  echo "== bench.py with the reference as the CPU baseline (cpu_baseline.kind)"
  python bench.py --no-configs --no-chain > gpurun_out/real_pygsp_bench.json 2> gpurun_out/real_pygsp_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/real_pygsp_bench.json").read().strip().splitlines()[-1])
print(json.dumps({"cpu_baseline": d.get("cpu_baseline"), "parity_vs_reference": d.get("parity_vs_reference"),
                  "parity_vs_oracle": d.get("parity_vs_oracle"), "value": d["value"], "roofline_frac": d["roofline"]["frac"]}, indent=1))
PY
} > gpurun_out/real_pygsp_gpu.log 2>&1
tail -60 gpurun_out/real_pygsp_gpu.log
