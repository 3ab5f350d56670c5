"""C-ABI checks that need no GPU: the library loads, exports exactly what include/gspx.h
declares, fails loudly without a device, and its step schedule is correct (executed on the CPU
by the oracle's plan runner)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, csr_from, rel_err
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine


def header_functions(names=("gspx.h", "gspx_ext.h")):
    """Functions declared in include/: gspx.h is the drop-in boundary of the path, gspx_ext.h the entry
    points beside it (opt-in evaluation, calibration)."""
    found = set()
    for name in names:
        src = open(os.path.join(ROOT, "include", name)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        found.update(re.findall(r"\b(gspx_[a-z0-9_]+)\s*\(", src))
    return sorted(found)


def exported(path):
    """Dynamic symbols gspx_* a shared object defines (nm -D)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("gspx_")})


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libgspx.so does not export " + n
    # the ctypes table and the headers agree
    assert sorted(_capi.SIGNATURES) == names
    # the boundary header stays what a maintainer binds for this one path: small
    core = header_functions(("gspx.h",))
    assert {"gspx_cheby_filter", "gspx_cheby_filter_dev", "gspx_graph_create_from_w", "gspx_graph_create_from_l",
            "gspx_graph_download_l", "gspx_gather", "gspx_comm_gather"} <= set(core)
    assert len(open(os.path.join(ROOT, "include", "gspx.h")).read().splitlines()) <= 220


def test_library_exports_exactly_the_headers():
    """One library, one header pair without feature macros (the experimental build was retired in round 6): what
    libgspx.so exports is exactly what include/*.h declares, and none of the retired kernels' entry points."""
    default_lib = os.path.join(ROOT, "pygsp_amd", "_lib", "libgspx.so")
    names = exported(default_lib)
    assert names == header_functions()
    assert not [n for n in names if "pair" in n or n in ("gspx_graph_set_tiles", "gspx_graph_tile_stats")]
    for h in ("gspx.h", "gspx_ext.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        assert "#ifdef GSPX_" not in src and "#if defined(GSPX_" not in src


def test_version_and_error_string():
    lib = _capi.load()
    assert b"gfx950" in lib.gspx_version()
    assert isinstance(_capi.last_error(), str)


@pytest.mark.skipif(_capi.device_count() > 0, reason="checks the no-device behaviour")
def test_fails_loudly_without_device():
    """No CPU fallback: without a HIP device the product path raises."""
    with pytest.raises(_capi.GspxError):
        engine.Context(0)


def test_argument_errors_need_no_device():
    lib = _capi.load()
    c = np.array([[1.0]])
    # M < 2 -> TypeError, as approximations.py:83-84
    with pytest.raises(TypeError):
        _capi.check(lib.gspx_plan_describe(None, 1, 1, _capi.ptr(c), None))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_plan_describe(None, 0, 3, _capi.ptr(c), None))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_cheby_filter_dev(None, 1.0, 1, 3, None, 1, None, None, 0, None))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_ctx_set_option(None, b"kernel", 1))
    n = ctypes.c_int64()
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_graph_n(None, ctypes.byref(n)))
    # the round-3 entry points refuse null handles / outputs the same way (also run under ASan, tests/test_asan.py)
    rep = np.zeros(12, dtype=np.int64)
    h = ctypes.c_void_p()
    ptr2 = np.array([0, 0], dtype=np.int32)
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_graph_setup(None, 1, 0, _capi.ptr(ptr2), None, None, _capi.F64, 0, _capi.F64, None, 0, 0, None,
                                         _capi.ptr(rep), ctypes.byref(h)))
    d4 = np.zeros(4)
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_graph_lmax_bounds(None, _capi.ptr(d4)))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_graph_download_perm(None, None))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_curve_order(None, 10, 2, _capi.ptr(np.zeros((10, 2))), 1, _capi.ptr(np.zeros(10, dtype=np.int32))))
    ms, gb = ctypes.c_double(), ctypes.c_double()
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_bench_gather(None, 1000, 64, 1000, 8, 1, 0.0, 8, 1, ctypes.byref(ms), ctypes.byref(gb)))
    d9 = (ctypes.c_double * 9)()
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_last_host_timing(None, d9))
    with pytest.raises(ValueError):
        _capi.check(lib.gspx_knn_search_stats(None, _capi.ptr(d4)))
    assert lib.gspx_comm_destroy(None) == 0


@pytest.mark.parametrize("order", [1, 2, 3, 4, 5, 6, 7, 30, 31, 32, 50])
def test_fused_schedule_reproduces_cheby_op(golden_sensor123, order):
    """The every-third-step flush schedule == the reference's per-step accumulation."""
    g = golden_sensor123
    L = csr_from(g, "Lcomb")
    lmax = float(g["lmax"])
    c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, order)
    plan = engine.plan_describe(c)
    assert plan.shape == (order, 7)
    # step 1 is T1 = 0.5 F T0, later steps T_k = F T_{k-1} - T_{k-2}
    assert plan[0, 0] == 0.5 and plan[0, 1] == 0.0
    assert np.all(plan[1:, 0] == 1.0) and np.all(plan[1:, 1] == -1.0)
    # exactly one final flush, on the last step; first flush writes, later ones accumulate
    assert plan[-1, 3] == 1 and plan[:-1, 3].sum() == 0
    fl = plan[:, 2][plan[:, 2] > 0]
    assert fl[0] == 1 and np.all(fl[1:] == 2)
    # at most ceil((K+1)/3) + 1 flushes: ~2/3 of an accumulator pass per step
    assert len(fl) <= (order + 1 + 2) // 3 + 1
    # every coefficient is used exactly once (c0 halved)
    used = plan[:, 4:].sum()
    assert abs(used - (c.sum() - 0.5 * c[0])) < 1e-12
    x = g["signals5"]
    y = orc.run_plan(L, lmax, plan, 1, x)[0]
    ref = orc.cheby_op(L, lmax, c, x)
    assert rel_err(y, ref) < 1e-13


def test_filterbank_schedule_is_deferred(golden_sensor123):
    lmax = float(golden_sensor123["lmax"])
    c = np.array([orc.compute_cheby_coeff(k, lmax, 20) for k in orc.mexican_hat_kernels(lmax, 4)])
    plan = engine.plan_describe(c)
    assert plan.shape == (20, 4 + 12)
    assert np.all(plan[:, 2] == 0)  # no in-step flush: all T_k kept, one combine pass at the end


def test_host_pipeline_schedules():
    """The batch schedule of the pipelined host-pointer call (gspx_host_pipeline_describe, host-only): batches add
    up to the panel, 128-byte rows with half-width first and last batches for large calls, the one-shot form for
    small ones, no ragged tail under 32-byte rows in the automatic schedule, explicit widths honoured."""
    import ctypes

    import numpy as np
    lib = _capi.load()

    def shape(mode, batch, edge, threads, dtype, N, nsig, planes=2):
        w = np.zeros(256, dtype=np.int64)
        n, t = ctypes.c_int(0), ctypes.c_int(0)
        _capi.check(lib.gspx_host_pipeline_describe(mode, batch, edge, threads, dtype, N, nsig, planes, _capi.ptr(w),
                                                    w.size, ctypes.byref(n), ctypes.byref(t)))
        return [int(v) for v in w[:n.value]], t.value

    assert shape(1, 0, 0, 0, _capi.F64, 1000000, 64)[0] == [8, 16, 16, 16, 8]      # the headline call
    assert shape(1, 0, 0, 0, _capi.F32, 1000000, 64)[0] == [16, 16, 16, 16]
    assert shape(1, 0, 0, 0, _capi.F32, 1000000, 256)[0] == [16] + [32] * 7 + [16]
    assert shape(1, 0, 0, 0, _capi.F64, 1000000, 4)[0] == []                       # too few columns: one shot
    assert shape(1, 0, 0, 0, _capi.F64, 20000, 64)[0] == []                        # under 48 MB of panels: one shot
    assert shape(0, 0, 0, 0, _capi.F64, 1000000, 64)[0] == []                      # switched off
    assert shape(2, 0, 0, 0, _capi.F64, 1000, 8)[0] == [4, 4]                      # "always": two halves at least
    assert shape(2, 0, 0, 0, _capi.F64, 1000, 2)[0] == []                          # (halves under 32-byte rows: merged)
    assert shape(2, 24, 0, 3, _capi.F64, 1000, 56) == ([24, 24, 8], 3)             # explicit width and threads
    assert shape(2, 16, 4, 0, _capi.F64, 1000, 40)[0] == [4, 16, 16, 4]
    for dtype, elt in ((_capi.F64, 8), (_capi.F32, 4)):
        for nsig in range(2, 200):
            for mode in (1, 2):
                w, t = shape(mode, 0, 0, 0, dtype, 3000000, nsig, planes=7)
                assert (not w) or (sum(w) == nsig and len(w) >= 2 and min(w) >= 1 and 1 <= t <= 64), (dtype, nsig, mode, w)
                if w and nsig * elt >= 64:
                    assert w[-1] * elt >= 32, (dtype, nsig, mode, w)
