# round 3: full GPU suite + the threaded / pipelined host paths under AddressSanitizer (host code instrumented)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
echo "asan runtime: $RT"
GSPX_LIB_PATH=$R/pygsp_amd/_lib/libgspx_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0 \
  timeout 900 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_2_kernels.py::test_host_pipeline_equals_one_shot_call tests/test_gpu_9_fuzz.py::test_soak_slice_of_the_host_pipeline tests/test_gpu_5_setup.py tests/test_gpu_6_multi.py::test_filter_columns_split_over_two_contexts tests/test_gpu_6_multi.py::test_gather_and_batch_across_contexts "tests/test_gpu_5_knn.py::test_highdim_golden" > $O/pytest_asan_gpu.log 2>&1; echo "asan pytest rc=$?"
grep -c "ERROR: AddressSanitizer" $O/pytest_asan_gpu.log; grep -m3 -A12 "ERROR: AddressSanitizer" $O/pytest_asan_gpu.log | head -40; tail -6 $O/pytest_asan_gpu.log
