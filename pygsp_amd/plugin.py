"""Drop-in for the real pygsp: route ``Filter.filter(method='chebyshev')`` through libgspx.

PyGSP has no plugin / FFI interface; the seam is Python attribute lookup:
``pygsp/filters/filter.py:309,319`` call ``approximations.cheby_op`` looked up on the module at
call time, and ``pygsp/filters/__init__.py:115`` re-exports it at import time.  ``install()``
replaces both with ``pygsp_amd.filters.cheby_op``; ``uninstall()`` restores the originals.

    import pygsp, pygsp_amd.plugin
    pygsp_amd.plugin.install()                  # from here on Chebyshev filtering runs on MI355X
    g = pygsp.filters.Heat(G, 50); y = g.filter(s, method='chebyshev', order=30)
"""
import numpy as np
from scipy import sparse

from . import engine, filters as _filters

_saved = {}
_config = {"laplacian": "device", "dtype": np.float64, "device": 0, "reorder": "auto", "tiles": "auto"}


def device_graph_for(G):
    """libgspx graph attached to a reference ``pygsp.graphs.Graph`` (cached on the object, rebuilt
    when ``G.compute_laplacian`` replaced ``G.L``, mirroring graph.py:602-609)."""
    key = (id(G.L), G.lap_type, np.dtype(_config["dtype"]).str, _config["laplacian"])
    cached = getattr(G, "_gspx_dev", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    ctx = engine.default_context(_config["device"])
    perm = None
    coords = getattr(G, "coords", None)
    if _config["reorder"] == "auto" and G.N >= 4096:
        perm = engine.auto_order(G.W, coords, _config["device"])
    elif _config["reorder"] == "rcm":
        perm = engine.locality_order(G.W, None)
    if _config["laplacian"] == "device":
        W = G.W if not G.is_directed() else sparse.csr_matrix((G.W + G.W.T) / 2)
        dev = engine.DeviceGraph.from_w(W, G.lap_type, dtype=_config["dtype"], perm=perm, ctx=ctx)
    else:  # bit-parity mode: upload the Laplacian the reference built
        dev = engine.DeviceGraph.from_l(G.L, dtype=_config["dtype"], perm=perm, ctx=ctx)
    if _config.get("tiles", "auto"):
        dev.auto_gather_tiles()
    G._gspx_dev = (key, dev)
    return dev


def install(pygsp_module=None, laplacian="device", dtype=np.float64, device=0, reorder="auto",
            tiles="auto"):
    """Patch the real pygsp in place.  `laplacian`: 'device' (L assembled by HIP kernels from
    G.W) or 'host' (upload the reference's G.L)."""
    if laplacian not in ("device", "host"):
        raise ValueError("laplacian must be 'device' or 'host'")
    if pygsp_module is None:
        import pygsp as pygsp_module
    _config.update(laplacian=laplacian, dtype=np.dtype(dtype), device=int(device), reorder=reorder,
                   tiles=tiles)
    approx = pygsp_module.filters.approximations
    if "cheby_op" not in _saved:
        _saved["cheby_op"] = approx.cheby_op
        _saved["alias"] = getattr(pygsp_module.filters, "cheby_op", None)
    approx.cheby_op = _filters.cheby_op
    pygsp_module.filters.cheby_op = _filters.cheby_op
    return pygsp_module


def uninstall(pygsp_module=None):
    if "cheby_op" not in _saved:
        return
    if pygsp_module is None:
        import pygsp as pygsp_module
    pygsp_module.filters.approximations.cheby_op = _saved.pop("cheby_op")
    alias = _saved.pop("alias")
    if alias is not None:
        pygsp_module.filters.cheby_op = alias
