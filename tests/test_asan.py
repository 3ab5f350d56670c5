"""The host side of libgspx under AddressSanitizer (a sanitizer build of the host
shim).  `make -C pygsp_amd/csrc asan` compiles the same source with -fsanitize=address for the host code
(about a minute, once; the device code is not instrumented) and the C-ABI checks that need no GPU -
argument validation, CSR validation, the schedule export executed against the oracle, error strings,
header = exports - run in a subprocess with the ASan runtime preloaded.  Any heap / stack / use-after-free
error in those paths aborts the subprocess."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asan_runtime():
    try:
        out = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                             text=True, timeout=60).stdout.strip()
    except Exception:
        return None
    if out and os.path.isabs(out) and os.path.exists(out):
        return out
    hits = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="needs hipcc")
def test_capi_host_paths_under_address_sanitizer():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no clang ASan runtime in this ROCm installation")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "pygsp_amd", "csrc"), "asan"],
                          stdout=subprocess.DEVNULL, env=dict(os.environ, HIPCC=HIPCC))
    lib = os.path.join(ROOT, "pygsp_amd", "_lib", "libgspx_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ, GSPX_LIB_PATH=lib, LD_PRELOAD=rt, PYTHONDONTWRITEBYTECODE="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1")
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                          os.path.join(ROOT, "tests", "test_capi.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = res.stdout[-2500:] + res.stderr[-2500:]
    assert res.returncode == 0, tail
    assert "AddressSanitizer" not in res.stderr, tail
    assert " passed" in res.stdout and "failed" not in res.stdout, tail
