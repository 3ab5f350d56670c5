"""Bindings of the EXPERIMENTAL build of libgspx (make -C pygsp_amd/csrc experimental -> _lib/libgspx_exp.so, selected
with GSPX_LIB_PATH): kernels that do not run by default because they measured slower (the fused Newton pair of
rounds 1-2) or are not cleared to run at size (two orders of the three-term recurrence per launch,
csrc/experimental/gspx_chebpair.hip.h; profiles/r04_pair_experiment.md).  Nothing in the product imports this
module; tools/pair_experiment.py and the opt-in tests do, and `attach()` refuses a default library.
"""
import ctypes

import numpy as np

from . import _capi
from .engine import DeviceGraph


def enable_pair_tiles(self):
    """Build (numpy) and upload the two-level tiles of the fused Newton-pair kernel.
    Returns the tile statistics.  A graph-setup step, ~seconds at N = 1M."""
    from . import tiling
    rp, col = self.download_internal()
    t = tiling.build_tiles(rp, col, self.N, 32)
    c = np.ascontiguousarray
    _capi.check(_capi.load().gspx_graph_set_tiles(
        self._h, 32, t["nb"], _capi.ptr(c(t["s1ptr"])), _capi.ptr(c(t["s1rows"])),
        _capi.ptr(c(t["s2ptr"])), _capi.ptr(c(t["s2rows"])), _capi.ptr(c(t["lidx1"])),
        _capi.ptr(c(t["occ_off"])), t["lidx2"].size, _capi.ptr(c(t["lidx2"])), t["max_n1"],
        t["max_n2"]))
    st = {k: t[k] for k in ("nb", "max_n1", "max_n2", "mean_n1", "mean_n2")}
    out = np.zeros(4, dtype=np.int64)
    _capi.check(_capi.load().gspx_graph_tile_stats(self._h, _capi.ptr(out)))
    st["unstaged_blocks"], st["fallback_lds_bytes"] = int(out[1]), int(out[2])
    return st


def enable_cheb_pair_tiles(self, block_rows=128):
    """Build (numpy, pygsp_amd/tiling.py) and upload the two-level tiles of the two-orders-per-launch
    recurrence kernel (gspx_graph_set_cheb_pair_tiles; opt-in experiment).  Returns the tile statistics.
    A graph set-up step: seconds at N = 1M."""
    from . import tiling
    rp, col = self.download_internal()
    t = tiling.build_tiles(rp, col, self.N, int(block_rows))
    c = np.ascontiguousarray
    stats = np.zeros(6, dtype=np.int64)
    _capi.check(_capi.load().gspx_graph_set_cheb_pair_tiles(
        self._h, int(block_rows), t["nb"], _capi.ptr(c(t["s1ptr"])), _capi.ptr(c(t["s1rows"])),
        _capi.ptr(c(t["s2ptr"])), _capi.ptr(c(t["s2rows"])), _capi.ptr(c(t["lidx1"])), _capi.ptr(c(t["occ_off"])),
        t["lidx2"].size, _capi.ptr(c(t["lidx2"])), _capi.ptr(stats)))
    return {"block_rows": int(block_rows), "nb": int(stats[0]), "max_n1": int(stats[1]), "max_n2": int(stats[2]),
            "max_entries_s1": int(stats[3]), "max_entries_own": int(stats[4]), "entries_level2": int(stats[5]),
            "mean_n1": t["mean_n1"], "mean_n2": t["mean_n2"]}


def disable_cheb_pair_tiles(self):
    _capi.check(_capi.load().gspx_graph_set_cheb_pair_tiles(self._h, 0, 0, None, None, None, None, None, None, 0,
                                                            None, None))


def cheby_pair_filter_dev(self, coeffs, x_ptr, y_ptr, nsig, lmax, chunk_lanes=4):
    """One filter of even order, two recurrence orders per launch (gspx_cheby_pair_filter_dev); device
    pointers in / out.  Returns device milliseconds of the whole call."""
    c = np.ascontiguousarray(np.asarray(coeffs, dtype=np.float64).reshape(-1))
    ms = ctypes.c_double(0)
    _capi.check(_capi.load().gspx_cheby_pair_filter_dev(
        self._h, float(lmax), c.size, _capi.ptr(c), int(nsig), ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr),
        int(chunk_lanes), ctypes.byref(ms)))
    return ms.value


def disable_pair_tiles(self):
    _capi.check(_capi.load().gspx_graph_set_tiles(self._h, 0, 0, None, None, None, None, None,
                                                  None, 0, None, 0, 0))


def attach():
    """Add the experimental methods to DeviceGraph.  Raises unless the loaded library is the experimental build."""
    _capi.load()
    if not _capi.experimental:
        raise _capi.GspxError(
            "the loaded libgspx ({}) is the default build: it does not export the experimental entry points. "
            "Build `make -C pygsp_amd/csrc experimental` and set GSPX_LIB_PATH={}".format(_capi.LIB_PATH,
                                                                                       _capi.EXP_LIB_PATH))
    DeviceGraph.enable_pair_tiles = enable_pair_tiles
    DeviceGraph.disable_pair_tiles = disable_pair_tiles
    DeviceGraph.enable_cheb_pair_tiles = enable_cheb_pair_tiles
    DeviceGraph.disable_cheb_pair_tiles = disable_cheb_pair_tiles
    DeviceGraph.cheby_pair_filter_dev = cheby_pair_filter_dev
