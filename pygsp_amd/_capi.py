"""ctypes binding of libgspx (C-ABI declared in include/gspx.h and include/gspx_ext.h).

The library is built in-tree (pygsp_amd/_lib/libgspx.so) by ``__graft_entry__.build()`` /
``make -C pygsp_amd/csrc``.  There is NO CPU fallback: if the shared object is missing, or no
HIP device is visible, every product entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GSPX_LIB_PATH: load another build of the same library instead, e.g. the sanitizer build of `make asan`)
LIB_PATH = os.environ.get("GSPX_LIB_PATH") or os.path.join(_HERE, "_lib", "libgspx.so")

F32, F64 = 0, 1
LAP_COMBINATORIAL, LAP_NORMALIZED = 0, 1
ANALYSIS, SYNTHESIS = 0, 1

OK, ERR_INVALID, ERR_COEFF, ERR_HIP, ERR_NODEVICE, ERR_OOM = 0, 1, 2, 3, 4, 5

_lib = None


class GspxError(RuntimeError):
    """HIP / runtime failure inside libgspx."""


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise ValueError("libgspx computes in float32 or float64, got {}".format(dt))


def np_dtype(code):
    return np.float32 if code == F32 else np.float64


# every exported symbol: name -> (restype, argtypes).  tests/test_capi.py checks this table
# against include/gspx.h + gspx_ext.h so the headers, the library and the binding cannot drift apart.
_c = ctypes
_P = _c.c_void_p
SIGNATURES = {
    "gspx_last_error": (_c.c_char_p, []),
    "gspx_version": (_c.c_char_p, []),
    "gspx_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "gspx_ctx_create": (_c.c_int, [_c.c_int, _c.POINTER(_P)]),
    "gspx_ctx_destroy": (_c.c_int, [_P]),
    "gspx_ctx_sync": (_c.c_int, [_P]),
    "gspx_ctx_set_option": (_c.c_int, [_P, _c.c_char_p, _c.c_int64]),
    "gspx_ctx_get_option": (_c.c_int, [_P, _c.c_char_p, _c.POINTER(_c.c_int64)]),
    "gspx_buf_alloc": (_c.c_int, [_P, _c.c_int64, _c.POINTER(_P)]),
    "gspx_buf_free": (_c.c_int, [_P]),
    "gspx_buf_upload": (_c.c_int, [_P, _P, _c.c_int64]),
    "gspx_buf_download": (_c.c_int, [_P, _P, _c.c_int64]),
    "gspx_buf_ptr": (_c.c_int, [_P, _c.POINTER(_P)]),
    "gspx_buf_bytes": (_c.c_int, [_P, _c.POINTER(_c.c_int64)]),
    "gspx_graph_create_from_w": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int,
                                            _c.c_int, _c.c_int, _P, _c.POINTER(_P)]),
    "gspx_graph_create_from_l": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int,
                                            _c.c_int, _P, _c.POINTER(_P)]),
    "gspx_graph_destroy": (_c.c_int, [_P]),
    "gspx_graph_n": (_c.c_int, [_P, _c.POINTER(_c.c_int64)]),
    "gspx_graph_nnz_l": (_c.c_int, [_P, _c.POINTER(_c.c_int64)]),
    "gspx_graph_nnz_internal": (_c.c_int, [_P, _c.POINTER(_c.c_int64)]),
    "gspx_graph_download_l": (_c.c_int, [_P, _P, _P, _P]),
    "gspx_graph_download_dw": (_c.c_int, [_P, _P]),
    "gspx_graph_build_ms": (_c.c_int, [_P, _c.POINTER(_c.c_double)]),
    "gspx_cheby_filter_dev": (_c.c_int, [_P, _c.c_double, _c.c_int, _c.c_int, _P, _c.c_int64, _P,
                                         _P, _c.c_int, _c.POINTER(_c.c_double)]),
    "gspx_cheby_filter": (_c.c_int, [_P, _c.c_double, _c.c_int, _c.c_int, _P, _c.c_int64, _P, _P,
                                     _c.c_int, _c.POINTER(_c.c_double)]),
    "gspx_newton_filter_dev": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _P, _c.c_int64, _P, _P,
                                          _c.POINTER(_c.c_double)]),
    "gspx_newton_filter": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _P, _c.c_int64, _P, _P,
                                      _c.POINTER(_c.c_double)]),
    "gspx_graph_download_internal": (_c.c_int, [_P, _P, _P]),
    "gspx_laplacian_apply_dev": (_c.c_int, [_P, _c.c_int64, _P, _P, _P]),
    "gspx_dirichlet_energy_dev": (_c.c_int, [_P, _c.c_int64, _P, _P, _P]),
    "gspx_tikhonov_cg_dev": (_c.c_int, [_P, _c.c_double, _P, _c.c_int64, _P, _P, _c.c_double,
                                         _c.c_double, _c.c_int64, _P, _P]),
    "gspx_graph_n_edges": (_c.c_int, [_P, _P]),
    "gspx_graph_download_edges": (_c.c_int, [_P, _P, _P, _P, _P, _P]),
    "gspx_graph_set_edge_list": (_c.c_int, [_P, _c.c_int64, _P, _P, _P, _c.c_int]),
    "gspx_grad_dev": (_c.c_int, [_P, _c.c_int64, _P, _P, _P]),
    "gspx_div_dev": (_c.c_int, [_P, _c.c_int64, _P, _P, _P]),
    "gspx_knn_build": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _c.c_int, _c.c_double, _c.c_int, _c.c_int, _P]),
    "gspx_knn_destroy": (_c.c_int, [_P]),
    "gspx_knn_info": (_c.c_int, [_P, _P, _P, _P]),
    "gspx_knn_download_w": (_c.c_int, [_P, _P, _P, _P]),
    "gspx_knn_search_stats": (_c.c_int, [_P, _P]),
    "gspx_knn_download_neighbors": (_c.c_int, [_P, _P, _P]),
    "gspx_graph_set_gather_tiles": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    "gspx_curve_keys": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _c.c_int, _P]),
    "gspx_graph_build_gather_tiles": (_c.c_int, [_P, _P]),
    "gspx_curve_order": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _c.c_int, _P]),
    "gspx_graph_setup": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int,
                                    _c.c_int, _P, _P, _c.POINTER(_P)]),
    "gspx_graph_setup_from_knn": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _P, _P, _c.POINTER(_P)]),
    "gspx_graph_download_perm": (_c.c_int, [_P, _P]),
    "gspx_graph_lmax_bounds": (_c.c_int, [_P, _P]),
    "gspx_sbm_build": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _P, _P, _c.c_uint64, _P]),
    "gspx_sbm_build_ex": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _P, _P, _c.c_uint64, _c.c_int, _P]),
    "gspx_radius_build": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _c.c_double, _c.c_double, _c.c_int, _P]),
    "gspx_last_timing": (_c.c_int, [_P, _c.POINTER(_c.c_double)]),
    "gspx_last_host_timing": (_c.c_int, [_P, _c.POINTER(_c.c_double)]),
    "gspx_last_host_timeline": (_c.c_int, [_P, _P, _c.c_int, _P]),
    "gspx_host_pipeline_describe": (_c.c_int, [_c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int64,
                                               _c.c_int64, _c.c_int, _P, _c.c_int, _P, _P]),
    "gspx_plan_describe": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _P]),
    "gspx_lanczos_lmax": (_c.c_int, [_P, _c.c_int, _c.c_double, _c.POINTER(_c.c_double),
                                     _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "gspx_bench_read": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.POINTER(_c.c_double)]),
    "gspx_gather": (_c.c_int, [_P, _c.c_int, _P, _P]),
    "gspx_identity_panel_dev": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _P]),
    "gspx_planes_pack_dev": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _c.c_int]),
    "gspx_device_pci_bus_id": (_c.c_int, [_c.c_int, _c.c_char_p, _c.c_int]),
    "gspx_comm_available": (_c.c_int, []),
    "gspx_comm_unique_id": (_c.c_int, [_P]),
    "gspx_comm_create": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _c.POINTER(_P)]),
    "gspx_comm_destroy": (_c.c_int, [_P]),
    "gspx_comm_info": (_c.c_int, [_P, _P]),
    "gspx_comm_gather": (_c.c_int, [_P, _P, _P, _c.c_int, _P, _c.POINTER(_c.c_double)]),
    "gspx_bench_gather": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_double, _c.c_int,
                                     _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "gspx_bench_copy": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.POINTER(_c.c_double)]),
    "gspx_poly_program_dev": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _P, _P, _c.c_int, _c.c_int64, _P, _P,
                                       _c.POINTER(_c.c_double)]),
    "gspx_poly_program": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _P, _P, _c.c_int, _c.c_int64, _P, _P,
                                   _c.POINTER(_c.c_double)]),
    "gspx_ctx_tune_placement": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _c.c_int64, _P, _P, _c.c_int, _c.c_int64, _P]),
    "gspx_bench_streams": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double)]),
    "gspx_bench_step_mix": (_c.c_int, [_P, _c.c_double, _c.c_int, _P, _c.c_int64, _P, _P, _c.c_int]),
}


def load():
    """Load libgspx.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GspxError(
            "libgspx.so not found at {} - build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C pygsp_amd/csrc`. pygsp_amd has no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().gspx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a libgspx status to the exception type the reference raises for the same condition."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)  # graph.py:635-639, filter.py:272-276
    if rc == ERR_COEFF:
        raise TypeError(msg)  # approximations.py:83-84
    if rc == ERR_NODEVICE:
        raise GspxError("no MI355X/HIP device: " + msg)
    if rc == ERR_OOM:
        raise GspxError("out of device memory: " + msg)
    raise GspxError(msg)


def ptr(arr):
    """Borrowed pointer to a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(ctypes.c_void_p)


def device_count():
    n = ctypes.c_int(0)
    rc = load().gspx_device_count(ctypes.byref(n))
    if rc != OK:
        return 0
    return n.value
