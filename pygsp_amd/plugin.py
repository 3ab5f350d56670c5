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

# The originals of everything install() replaces are kept ON the patched object itself (attribute _gspx_saved of the
# approximations module, of the Filter class, of the Graph class), so that two pygsp-shaped modules can be patched
# and restored independently (one global table restored the first module's functions into the second).
_SAVED = "_gspx_saved"
_installed = {}  # id(module) -> module, in installation order (uninstall() without an argument: the last one)
_config = {"laplacian": "device", "dtype": np.float64, "device": 0, "reorder": "auto", "tiles": "auto",
           "devices": None, "evaluation": None}
_cache_lock = threading.RLock()


def _finite_coords(G):
    coords = getattr(G, "coords", None)
    try:
        coords = np.asarray(coords, dtype=np.float64)
    except (TypeError, ValueError):
        return None
    if coords.ndim != 2 or coords.shape[0] != G.N or coords.shape[1] < 2 or not np.isfinite(coords).all():
        return None  # plotting layouts may hold NaN / inf, a 1-D cloud has no curve: the pattern-based order
    return coords


def device_graph_for(G, ctx=None, dtype=None):
    """libgspx graph attached to a reference ``pygsp.graphs.Graph``: cached on the object next to the
    very ``G.L`` (and ``G.W``) it was built from, and rebuilt when ``G.compute_laplacian`` replaced
    them, mirroring graph.py:602-609.  The cache entry holds the matrices themselves and compares with
    ``is`` - an ``id()`` alone could be recycled by a later matrix.  One entry per context (`ctx`: the
    context of the configured device unless given - the replicas of a device list, pygsp_amd.multi).
    A stale entry is dropped, not destroyed: whoever still holds the old DeviceGraph (another thread in the
    middle of a filter call, a caller that kept the return value) keeps a live handle, and the device memory
    goes when the last reference does (DeviceGraph.__del__)."""
    dtype = np.dtype(dtype or _config["dtype"])
    conf = (G.lap_type, dtype.str, _config["laplacian"], _config["reorder"], bool(_config.get("tiles", "auto")))
    ctx = ctx or engine.default_context(_config["device"])
    key = (id(ctx), dtype.str)
    with _cache_lock:
        cache = G.__dict__.setdefault("_gspx_dev", {})
        cached = cache.get(key)
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
            # checks, directedness, (W + W.T) / 2 of a directed graph, vertex order and Laplacian in one device call on
            # the uploaded G.W (gspx_graph_setup)
            try:
                dev, _ = engine.DeviceGraph.setup(G.W, G.lap_type, dtype, coords, order, ctx=ctx)
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
                dev = engine.DeviceGraph.from_w(W, G.lap_type, dtype=dtype, perm=perm, ctx=ctx)
            else:  # bit-parity mode: upload the Laplacian the reference built
                dev = engine.DeviceGraph.from_l(G.L, dtype=dtype, perm=perm, ctx=ctx)
        if _config.get("tiles", "auto"):
            dev.auto_gather_tiles()
        cache[key] = (G.L, G.W, conf, dev, ctx)
        return dev


def _estimate_lmax_on_device(self, method="lanczos"):
    """``Graph.estimate_lmax`` of the reference (graph.py:858-931) with its 'lanczos' branch on the device: the
    ARPACK call of graph.py:911-917 (3.3 s at N = 1M, random start vector) becomes gspx_lanczos_lmax on the
    graph's device Laplacian (milliseconds, deterministic).  Same contract - a Ritz value from below, checked
    against the upper bound, increased by 1 % (graph.py:919-920), ValueError when it does not converge - and the
    same caching through ``_lmax_method``; every other method is the reference's own code."""
    if method != "lanczos":
        return _saved_on(type(self))["estimate_lmax"](self, method)
    if method == self._lmax_method:
        return
    ritz = _lanczos_ritz(device_graph_for(self, dtype=np.float64))
    if not ritz <= self._get_upper_bound() * (1 + 1e-6) + 1e-12:  # graph.py:919 (an assert there)
        raise ValueError("The Lanczos estimate {} exceeds the upper bound of lambda_max.".format(ritz))
    self._lmax_method = method
    self._lmax = ritz * 1.01


def _lanczos_ritz(dev):
    """Largest Ritz value of the float64 device Laplacian; a second, longer run before giving up (the
    reference's ARPACK call stops at tol 5e-3, graph.py:911-917: where it succeeds this must not fail)."""
    try:
        return dev.lanczos_lmax(max_iter=80, tol=5e-4)[0]
    except ValueError:
        return dev.lanczos_lmax(max_iter=400, tol=5e-3)[0]


def to_device(G, s):
    """Upload a signal of a reference ``pygsp.graphs.Graph`` once and keep it on the device (engine.DeviceArray):
    with install(wrap_filter=True) the real ``Filter.filter`` / ``analyze`` / ``synthesize`` take it and return
    one, so a chain of filters costs one upload and one download (``np.asarray(result)``)."""
    arr = G._check_signal(s)
    dev = device_graph_for(G)
    return engine.DeviceArray.from_host(dev.ctx, arr, dev.dtype)


def tune_placement(bank, n_signals, order=30, candidates=32, stride_mb=8000):
    """Placement tuning (filters.tune_placement) for a real ``pygsp.filters.Filter`` of one kernel after install():
    its own compute_cheby_coeff, the graph's device Laplacian on the configured context."""
    return _filters.tune_placement(bank, n_signals, order, candidates, _reference_coefficients, stride_mb)


def _saved_on(cls):
    """The originals saved on the patched class `cls` derives from (or is)."""
    for k in cls.__mro__:
        saved = k.__dict__.get(_SAVED)
        if saved is not None:
            return saved
    raise RuntimeError("pygsp_amd.plugin: {} is not a patched class".format(cls.__name__))


def _reference_coefficients(bank, m):
    # the coefficients of the bank's own package (approximations.py:9-55): its own code, its own kernels
    return _saved_on(type(bank))["approximations"].compute_cheby_coeff(bank, m=m)


def _filter_on_device(self, s, method="chebyshev", order=30):
    """``pygsp.filters.Filter.filter`` (filter.py:146-328) with its Chebyshev branch in one device call: the shape
    rules are the reference's (filters._cube_shape restates filter.py:267-290), analysis is the call the
    reference makes, synthesis is ONE vector-coefficient recurrence on the device instead of the loop of
    filter.py:318-321 (Nf cheby_op calls, each with its own host round trip), and a DeviceArray stays on the
    device.  Everything else (method='exact', unknown methods) is the reference's own code."""
    if method != "chebyshev":
        return _saved_on(type(self))["filter"](self, s, method=method, order=order)
    return _filters.filter_signals(self, s, method, order, None, _reference_coefficients)


def _compute_frame_on_device(self, **kwargs):
    """``Filter.compute_frame`` (filter.py:506-600) without its N x N host identity: the deltas are written on
    the device panel by panel (filters.frame_panels); same (Nf N, N) result.  Other methods, and graphs whose
    size collides with the shape rules (N in {1, Nf}: the identity then reads as a synthesis input,
    filter.py:270), run the reference's own code."""
    extra = set(kwargs) - {"method", "order"}
    if kwargs.get("method", "chebyshev") != "chebyshev" or extra or self.G.N in (1, self.Nf):
        return _saved_on(type(self))["compute_frame"](self, **kwargs)
    if self.G.N > 2000:  # filter.py:593-596
        _filters_logger(self).warning("Creating a big matrix. You should prefer the filter method.")
    return _filters.frame_panels(self, kwargs.get("order", 30), coefficients=_reference_coefficients)


def _filters_logger(bank):
    import logging
    return logging.getLogger(type(bank).__module__)


def install(pygsp_module=None, laplacian="device", dtype=np.float64, device=0, reorder="auto",
            tiles="auto", devices=None, lmax="reference", wrap_filter=True, evaluation=None):
    """Patch the real pygsp in place.  `laplacian`: 'device' (L assembled by HIP kernels from
    G.W) or 'host' (upload the reference's G.L).  `devices` (a list of GPU ids, optional): every
    ``Filter.filter(method='chebyshev')`` splits its signal columns over these GPUs - the graph is replicated
    once per GPU, the outputs are gathered by RCCL inside libgspx (pygsp_amd.multi.filter_columns).
    `lmax`: 'device' also replaces ``Graph.estimate_lmax`` so that its default 'lanczos' method runs on the device
    (the step right before the path); 'reference' (default) leaves ARPACK in place.
    `wrap_filter` (default True; the secondary seam of SURVEY 8(b)): also replace ``Filter.filter`` and
    ``Filter.compute_frame`` (filter.py:146, 506) - fused synthesis, device-built identity panels,
    device-resident arrays; False patches ``cheby_op`` alone, and the reference's own loops call it.
    `evaluation`: how the polynomial of a SINGLE filter is evaluated in analysis calls - 'recurrence' (the
    reference's three-term recurrence, approximations.py:99-112), 'newton' (the same polynomial in Newton form:
    fewer panel passes per order) or 'auto' (Newton where it is faster and filters.newton_guard() clears the
    polynomial for the compute dtype, the recurrence otherwise); None (default): pygsp_amd.filters.EVALUATION,
    which is 'recurrence' unless filters.set_evaluation() changed it."""
    if evaluation is not None and evaluation not in _filters.EVALUATIONS:
        raise ValueError("evaluation must be 'recurrence', 'newton', 'product' or 'auto'")
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
                   tiles=tiles, devices=devices, evaluation=evaluation)
    approx = pygsp_module.filters.approximations
    if _SAVED not in approx.__dict__:
        setattr(approx, _SAVED, {"cheby_op": approx.cheby_op, "alias": getattr(pygsp_module.filters, "cheby_op", None)})
    _installed.pop(id(pygsp_module), None)
    _installed[id(pygsp_module)] = pygsp_module
    approx.cheby_op = _filters.cheby_op
    pygsp_module.filters.cheby_op = _filters.cheby_op
    filter_cls = getattr(pygsp_module.filters, "Filter", None)
    can_wrap = filter_cls is not None and hasattr(approx, "compute_cheby_coeff")
    if wrap_filter and can_wrap:
        if _SAVED not in filter_cls.__dict__:
            setattr(filter_cls, _SAVED, {"filter": filter_cls.filter, "compute_frame": filter_cls.compute_frame,
                                         "approximations": approx})
        filter_cls.filter, filter_cls.compute_frame = _filter_on_device, _compute_frame_on_device
    elif filter_cls is not None:
        _restore(filter_cls, ("filter", "compute_frame"))
    graph_cls = getattr(getattr(pygsp_module, "graphs", None), "Graph", None)
    if lmax == "device" and graph_cls is not None:
        if _SAVED not in graph_cls.__dict__:
            setattr(graph_cls, _SAVED, {"estimate_lmax": graph_cls.estimate_lmax})
        graph_cls.estimate_lmax = _estimate_lmax_on_device
    elif graph_cls is not None:
        _restore(graph_cls, ("estimate_lmax",))
    return pygsp_module


def _restore(obj, names):
    """Put back the originals saved on `obj` (a class or a module) and drop the table; no-op when there is none."""
    saved = obj.__dict__.get(_SAVED)
    if saved is None:
        return None
    for n in names:
        setattr(obj, n, saved[n])
    delattr(obj, _SAVED)
    return saved


def uninstall(pygsp_module=None):
    """Restore what install() replaced in `pygsp_module` (default: the module patched last; the importable pygsp
    when nothing is recorded).  Other patched modules stay patched."""
    if pygsp_module is None:
        if _installed:
            pygsp_module = list(_installed.values())[-1]
        else:
            try:
                import pygsp as pygsp_module
            except ImportError:
                return
    _installed.pop(id(pygsp_module), None)
    if not _installed:
        _config["evaluation"] = None
    approx = pygsp_module.filters.approximations
    saved = _restore(approx, ("cheby_op",))
    if saved is None:
        return
    if saved["alias"] is not None:
        pygsp_module.filters.cheby_op = saved["alias"]
    filter_cls = getattr(pygsp_module.filters, "Filter", None)
    if filter_cls is not None:
        _restore(filter_cls, ("filter", "compute_frame"))
    graph_cls = getattr(getattr(pygsp_module, "graphs", None), "Graph", None)
    if graph_cls is not None:
        _restore(graph_cls, ("estimate_lmax",))


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
