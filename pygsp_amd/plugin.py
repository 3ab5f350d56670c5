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

import threading

_saved = {}
_config = {"laplacian": "device", "dtype": np.float64, "device": 0, "reorder": "auto", "tiles": "auto",
           "devices": None}
_cache_lock = threading.RLock()


def _finite_coords(G):
    coords = getattr(G, "coords", None)
    try:
        coords = np.asarray(coords, dtype=np.float64)
    except (TypeError, ValueError):
        return None
    if coords.ndim != 2 or coords.shape[0] != G.N or not np.isfinite(coords).all():
        return None  # plotting layouts may hold NaN / inf: fall back to the pattern-based order
    return coords


def device_graph_for(G, ctx=None):
    """libgspx graph attached to a reference ``pygsp.graphs.Graph``: cached on the object next to the
    very ``G.L`` (and ``G.W``) it was built from, and rebuilt when ``G.compute_laplacian`` replaced
    them, mirroring graph.py:602-609.  The cache entry holds the matrices themselves and compares with
    ``is`` - an ``id()`` alone could be recycled by a later matrix.  One entry per context (`ctx`: the
    context of the configured device unless given - the replicas of a device list, pygsp_amd.multi).
    A stale entry is dropped, not destroyed: whoever still holds the old DeviceGraph (another thread in the
    middle of a filter call, a caller that kept the return value) keeps a live handle, and the device memory
    goes when the last reference does (DeviceGraph.__del__)."""
    conf = (G.lap_type, np.dtype(_config["dtype"]).str, _config["laplacian"], _config["reorder"],
            bool(_config.get("tiles", "auto")))
    ctx = ctx or engine.default_context(_config["device"])
    with _cache_lock:
        cache = G.__dict__.setdefault("_gspx_dev", {})
        cached = cache.get(id(ctx))
        if (cached is not None and cached[0] is G.L and cached[1] is G.W and cached[2] == conf
                and cached[4] is ctx and getattr(cached[3], "_h", None)):
            return cached[3]
        dev = None
        coords = _finite_coords(G)
        order = _config["reorder"]
        if order == "auto" and coords is None and G.N >= 4096 and sparse.isspmatrix_csr(G.W) and engine.expander_like(G.W):
            order = "none"  # a random-like graph (ER / SBM): no order helps, skip reverse Cuthill-McKee
        one_call = (_config["laplacian"] == "device" and sparse.isspmatrix_csr(G.W) and order in ("auto", "none", None, False)
                    and not (order == "auto" and coords is None and G.N >= 4096))  # (no coordinates: RCM is host work)
        if one_call:
            # checks, directedness, vertex order and Laplacian in one device call on the uploaded G.W
            # (gspx_graph_setup); a directed graph / explicit zeros come back as None and take the route below
            try:
                dev, _ = engine.DeviceGraph.setup(G.W, G.lap_type, _config["dtype"], coords, order, ctx=ctx)
            except ValueError:
                dev = None
        if dev is None:
            perm = None
            if order == "auto" and G.N >= 4096:
                perm = engine.auto_order(G.W, coords, ctx=ctx)
            elif order == "rcm":
                perm = engine.locality_order(G.W, None)
            if _config["laplacian"] == "device":
                W = G.W if not G.is_directed() else sparse.csr_matrix((G.W + G.W.T) / 2)
                dev = engine.DeviceGraph.from_w(W, G.lap_type, dtype=_config["dtype"], perm=perm, ctx=ctx)
            else:  # bit-parity mode: upload the Laplacian the reference built
                dev = engine.DeviceGraph.from_l(G.L, dtype=_config["dtype"], perm=perm, ctx=ctx)
        if _config.get("tiles", "auto"):
            dev.auto_gather_tiles()
        cache[id(ctx)] = (G.L, G.W, conf, dev, ctx)
        return dev


def _estimate_lmax_on_device(self, method="lanczos"):
    """``Graph.estimate_lmax`` of the reference (graph.py:858-931) with its 'lanczos' branch on the device: the
    ARPACK call of graph.py:911-917 (3.3 s at N = 1M, random start vector) becomes gspx_lanczos_lmax on the
    graph's device Laplacian (milliseconds, deterministic).  Same contract - a Ritz value from below, checked
    against the upper bound, increased by 1 % (graph.py:919-920), ValueError when it does not converge - and the
    same caching through ``_lmax_method``; every other method is the reference's own code."""
    if method != "lanczos":
        return _saved["estimate_lmax"](self, method)
    if method == self._lmax_method:
        return
    ritz, _ = device_graph_for(self).lanczos_lmax(max_iter=80, tol=5e-4)
    assert ritz <= self._get_upper_bound() * (1 + 1e-6) + 1e-12
    self._lmax_method = method
    self._lmax = ritz * 1.01


def install(pygsp_module=None, laplacian="device", dtype=np.float64, device=0, reorder="auto",
            tiles="auto", devices=None, lmax="reference"):
    """Patch the real pygsp in place.  `laplacian`: 'device' (L assembled by HIP kernels from
    G.W) or 'host' (upload the reference's G.L).  `devices` (a list of GPU ids, optional): every
    ``Filter.filter(method='chebyshev')`` splits its signal columns over these GPUs - the graph is replicated
    once per GPU, the outputs are gathered by RCCL inside libgspx (pygsp_amd.multi.filter_columns).
    `lmax`: 'device' also replaces ``Graph.estimate_lmax`` so that its default 'lanczos' method runs on the device
    (the step right before the path); 'reference' (default) leaves ARPACK in place."""
    if laplacian not in ("device", "host"):
        raise ValueError("laplacian must be 'device' or 'host'")
    if lmax not in ("device", "reference"):
        raise ValueError("lmax must be 'device' or 'reference'")
    if pygsp_module is None:
        import pygsp as pygsp_module
    if devices is not None:
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        device = devices[0]
    _config.update(laplacian=laplacian, dtype=np.dtype(dtype), device=int(device), reorder=reorder,
                   tiles=tiles, devices=devices)
    approx = pygsp_module.filters.approximations
    if "cheby_op" not in _saved:
        _saved["cheby_op"] = approx.cheby_op
        _saved["alias"] = getattr(pygsp_module.filters, "cheby_op", None)
    approx.cheby_op = _filters.cheby_op
    pygsp_module.filters.cheby_op = _filters.cheby_op
    graph_cls = getattr(getattr(pygsp_module, "graphs", None), "Graph", None)
    if lmax == "device" and graph_cls is not None:
        if "estimate_lmax" not in _saved:
            _saved["estimate_lmax"] = graph_cls.estimate_lmax
        graph_cls.estimate_lmax = _estimate_lmax_on_device
    elif "estimate_lmax" in _saved and graph_cls is not None:
        graph_cls.estimate_lmax = _saved.pop("estimate_lmax")
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
    graph_cls = getattr(getattr(pygsp_module, "graphs", None), "Graph", None)
    if "estimate_lmax" in _saved and graph_cls is not None:
        graph_cls.estimate_lmax = _saved.pop("estimate_lmax")


def use_backend(name, pygsp_module=None, **install_options):
    """Select who evaluates ``Filter.filter(method='chebyshev')`` of the real pygsp: 'gspx' (this
    engine, = install()) or 'reference' (scipy on the host, = uninstall())."""
    if name == "gspx":
        return install(pygsp_module, **install_options)
    if name == "reference":
        return uninstall(pygsp_module)
    raise ValueError("backend must be 'gspx' or 'reference', got {!r}".format(name))


def _backend_from_env():
    """PYGSP_AMD_BACKEND=gspx in the environment installs the seam as soon as this module is imported
    (e.g. from a sitecustomize / conftest line ``import pygsp_amd.plugin``); anything else: no-op."""
    import os
    if os.environ.get("PYGSP_AMD_BACKEND", "").strip().lower() == "gspx":
        try:
            import pygsp
        except ImportError:
            return
        install(pygsp)


_backend_from_env()
