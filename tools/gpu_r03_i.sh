# round 3: full GPU suite
# (the host-sanitizer build cannot run here: the HIP runtime aborts at device initialisation under ASan on this
# stack, so tests/test_asan.py covers the host paths that need no device)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
