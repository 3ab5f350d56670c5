"""pygsp.graphs.Graph surface for the Chebyshev-filtering hot path, backed by libgspx.

Mirrors (same names, argument meaning and exceptions) the parts of
``pygsp/graphs/graph.py`` the path touches:

* ``Graph.__init__``              graph.py:98-176   (validation, n_vertices/N, n_edges/Ne)
* ``Graph.compute_laplacian``     graph.py:510-630  -> built ON DEVICE (gspx_graph_create_from_w)
* ``Graph.dw``                    graph.py:783-838  -> device column sums
* ``Graph.lmax / estimate_lmax``  graph.py:840-960  (host, scipy: a read-only input of the path)
* ``Graph._check_signal``         graph.py:632-640
* ``Graph.is_directed``           graph.py:357-405

Everything else of the reference's Graph (plotting, I/O, subgraphs, ...) is out of scope; use
the real pygsp together with ``pygsp_amd.plugin.install()`` for those.

The synthetic generators at the bottom (Sensor, ErdosRenyi, StochasticBlockModel) are vectorised
samplers of the same distributions as the reference's constructors (the reference's ER/SBM
constructors are O(N^2) Python loops, stochasticblockmodel.py:125-139, unusable at N >= 1e5).
"""
import logging

import numpy as np
from scipy import sparse, spatial
from scipy.sparse import linalg as splinalg

from . import engine

_logger = logging.getLogger(__name__)


def _checked_adjacency(adjacency, log):
    """The weight matrix as canonical-ready CSR, after the reference's checks (graph.py:98-134): square,
    finite; self-loops and negative weights only warn."""
    dense = None if sparse.issparse(adjacency) else np.asanyarray(adjacency)
    shape = adjacency.shape if dense is None else dense.shape
    if len(shape) != 2 or shape[0] != shape[1]:
        raise ValueError("Adjacency: must be a square matrix.")
    W = sparse.csr_matrix(adjacency if dense is None else dense, copy=False)
    mass = W.sum()
    if np.isnan(mass):
        raise ValueError("Adjacency: there is a Not a Number (NaN).")
    if np.isinf(mass):
        raise ValueError("Adjacency: there is an infinite value.")
    if np.any(W.diagonal() != 0):
        log.warning("Adjacency: there are self-loops (non-zeros on the diagonal). "
                    "The Laplacian will not see them.")
    if W.nnz and W.data.min() < 0:
        log.warning("Adjacency: there are negative edge weights.")
    W.eliminate_zeros()
    return W


class Graph:
    """A weighted graph whose Laplacian lives on the GPU.  Same constructor contract as
    pygsp.graphs.Graph for the path (adjacency, lap_type, coords, plotting); the keyword-only arguments
    are this engine's: compute dtype, device, internal vertex order, gather tiles."""

    def __init__(self, adjacency, lap_type="combinatorial", coords=None, plotting={}, *,
                 compute_dtype=np.float64, device=0, reorder="auto", tiles="auto", ctx=None):
        self.logger = _logger
        # engine-side configuration and state
        # (ctx: a specific libgspx context instead of the default one of `device` - one per driver thread
        # when a process runs several GPUs, pygsp_amd.multi)
        self._ctx = ctx
        self.compute_dtype, self.device = np.dtype(compute_dtype), int(ctx.device if ctx is not None else device)
        self.reorder = reorder
        self.tiles = tiles  # "auto" | True | False: gather tiles of the LDS-staged recurrence step
        self.tile_stats = None
        self._perm, self._perm_done = None, False
        self._dev = {}     # compute dtype -> engine.DeviceGraph
        self._adj_host = self._adj_dev = None  # W: the scipy matrix / a builder's result still on the device
        self._L = self._dw = None
        self._flags = {}   # lazily computed facts about W ("directed")
        self._forget_spectrum()
        if lap_type not in ("combinatorial", "normalized"):
            raise ValueError("Unknown Laplacian type {}".format(lap_type))
        self.lap_type = lap_type
        if coords is not None:
            self.coords = np.asanyarray(coords)
        self.plotting, self.signals = dict(plotting), {}
        if isinstance(adjacency, engine.DeviceAdjacency) and not self._setup_from_device_adjacency(adjacency):
            adjacency = adjacency.download()  # (a directed symmetrisation type: the host route below)
        if self._adj_dev is None and not self._setup_on_device(adjacency):
            # the host route (dense input, a directed graph, explicit zeros, ...): the reference's steps one by one
            self._adjacency = _checked_adjacency(adjacency, self.logger)
            self.n_vertices = self.N = self._adjacency.shape[0]
            # stored entries of a directed graph; unordered pairs (+ self-loops) of an undirected one
            loops = int(np.count_nonzero(self._adjacency.diagonal()))
            stored = self._adjacency.nnz
            self.n_edges = self.Ne = stored if self.is_directed() else (stored - loops) // 2 + loops
            self.compute_laplacian(lap_type)

    def _setup_on_device(self, adjacency):
        """Fast route for a scipy CSR matrix: ONE device call (engine.DeviceGraph.setup -> gspx_graph_setup) does the
        checks of graph.py:98-134, the directedness test of graph.py:357-405, the internal vertex order and the
        Laplacian of graph.py:510-630 on the uploaded matrix - no host pass over the stored entries.  False when
        the host route has to take over."""
        if not sparse.isspmatrix_csr(adjacency) or adjacency.shape[0] != adjacency.shape[1]:
            return False
        order = self.reorder
        if isinstance(order, str) and order == "rcm":
            return False
        coords = self._curve_coords(adjacency.shape[0])
        if order == "auto" and coords is None and adjacency.shape[0] >= 4096:
            # no coordinates: the candidate order is reverse Cuthill-McKee, a host (scipy) algorithm - unless the graph
            # is random-like (ER / SBM: balls grow by the mean degree per hop), where no order helps
            if not engine.expander_like(adjacency):
                return False
            order = "none"
        if order in ("morton", "hilbert") and adjacency.shape[0] < 4096:
            return False  # an explicitly requested curve on a small graph: the numpy curves of engine.locality_order
        try:
            dev, rep = engine.DeviceGraph.setup(adjacency, self.lap_type, self.compute_dtype, coords, order,
                                                ctx=self.context)
        except ValueError as e:
            if "canonical" in str(e):
                return False  # duplicates / unsorted indices: the host route sums and sorts them
            raise
        if dev is None:  # (a directed graph whose union pattern would not fit 32-bit indices)
            return False
        if rep["self_loops"]:
            self.logger.warning("Adjacency: there are self-loops (non-zeros on the diagonal). "
                                "The Laplacian will not see them.")
        if rep["negative"]:
            self.logger.warning("Adjacency: there are negative edge weights.")
        if rep["zeros"]:  # graph.py:126-128: W keeps no edge of weight 0 (in place, like the reference; the device
            adjacency.eliminate_zeros()  # Laplacian was built without them)
        self._adjacency = adjacency
        self.n_vertices = self.N = adjacency.shape[0]
        # directed (W - W.T has a nonzero, counted on the device): the Laplacian was built from (W + W.T) / 2 in the
        # same device call (graph.py:613-616); n_edges: stored entries, or unordered pairs + self-loops (graph.py:135-140)
        directed = self._flags["directed"] = rep["asymmetric"] > 0
        stored, loops = adjacency.nnz, rep["self_loops"]
        self.n_edges = self.Ne = stored if directed else (stored - loops) // 2 + loops
        self._perm_done, self._perm_lazy = True, dev if rep["reordered"] else None
        self.setup_report = rep
        if self.tiles == "auto":
            self.tile_stats = dev.auto_gather_tiles()
        elif self.tiles:
            self.tile_stats = dev.enable_gather_tiles()
        self._dev[self.compute_dtype] = dev
        return True

    def _setup_from_device_adjacency(self, adjacency):
        """A generator class hands over the W its device builder produced (engine.DeviceAdjacency): checks, vertex
        order and Laplacian run on it where it lies (gspx_graph_setup_from_knn); the host copy behind G.W is made
        when somebody asks for it.  False when the host route has to take over."""
        order = self.reorder
        if isinstance(order, str) and order == "rcm":
            return False
        coords = self._curve_coords(adjacency.shape[0])
        if order == "auto" and coords is None:
            if getattr(self, "coords", None) is not None and adjacency.shape[0] >= 4096:
                return False  # coordinates no curve can use (a 1-D point cloud): the host route tries RCM
            order = "none"  # a sampled block model: no coordinates, and no order helps a random graph
        if order in ("morton", "hilbert") and adjacency.shape[0] < 4096:
            return False
        dev, rep = engine.DeviceGraph.setup_from(adjacency, self.lap_type, self.compute_dtype, coords, order)
        if dev is None or rep["zeros"]:  # (explicit zeros: W itself has to lose them - the host route)
            if dev is not None:
                dev.destroy()
            return False
        if rep["self_loops"]:
            self.logger.warning("Adjacency: there are self-loops (non-zeros on the diagonal). "
                                "The Laplacian will not see them.")
        self._adj_dev = adjacency
        self.n_vertices = self.N = adjacency.shape[0]
        directed = self._flags["directed"] = rep["asymmetric"] > 0  # (a 'tril' / 'triu' neighbour graph, a directed block model)
        stored, loops = adjacency.nnz, rep["self_loops"]
        self.n_edges = self.Ne = stored if directed else (stored - loops) // 2 + loops
        self._perm_done, self._perm_lazy = True, dev if rep["reordered"] else None
        self.setup_report = rep
        if self.tiles == "auto":
            self.tile_stats = dev.auto_gather_tiles()
        elif self.tiles:
            self.tile_stats = dev.enable_gather_tiles()
        self._dev[self.compute_dtype] = dev
        return True

    def _curve_coords(self, n_vertices):
        """The coordinates a space-filling curve can order (the test of engine.DeviceGraph._order_args: one row per
        vertex, at least two columns), else None - a 1-D point cloud or a layout of another size then takes the
        same route as a graph without coordinates (random-like: no order; otherwise reverse Cuthill-McKee on the
        host) instead of silently keeping the graph's own order."""
        coords = getattr(self, "coords", None)
        ok = coords is not None and np.ndim(coords) == 2 and np.shape(coords)[0] == n_vertices and np.shape(coords)[1] >= 2
        return coords if ok else None

    @property
    def _adjacency(self):
        if self._adj_host is None and self._adj_dev is not None:
            self._adj_host, self._adj_dev = self._adj_dev.download(), None
        return self._adj_host

    @_adjacency.setter
    def _adjacency(self, W):
        self._adj_host, self._adj_dev = W, None

    def _forget_spectrum(self):
        """Everything derived from the Laplacian's spectrum (graph.py:602-609)."""
        self._lmax = self._lmax_method = self._e = self._U = None

    # ---- adjacency -----------------------------------------------------------------------------
    @property
    def W(self):
        return self._adjacency

    def is_directed(self):
        """True when W differs from its transpose (graph.py:357-405); computed once."""
        if "directed" not in self._flags:
            W = self._adjacency
            self._flags["directed"] = bool((W - W.T).count_nonzero())
        return self._flags["directed"]

    @property
    def _directed(self):
        return self._flags.get("directed")

    def is_connected(self):
        """graph.py:303-355: every vertex reachable from every other one - for a directed graph along the
        edges AND against them (strongly connected).  Connected components of W on the host (scipy's csgraph),
        computed once."""
        if "connected" not in self._flags:
            from scipy.sparse import csgraph
            if self.n_vertices == 0:
                self._flags["connected"] = True
            else:
                n, _ = csgraph.connected_components(self.W, directed=self.is_directed(), connection="strong",
                                                    return_labels=True)
                self._flags["connected"] = bool(n == 1)
        return self._flags["connected"]

    def _symmetric_w(self):
        """W itself if undirected, else (W + W.T)/2 (utils.symmetrize 'average', graph.py:613-616)."""
        return sparse.csr_matrix((self.W + self.W.T) / 2) if self.is_directed() else self.W

    @property
    def context(self):
        """The libgspx context (device + stream) this graph's device state lives on."""
        return self._ctx if self._ctx is not None else engine.default_context(self.device)

    # ---- Laplacian (device) --------------------------------------------------------------------
    def _internal_order(self):
        lazy = getattr(self, "_perm_lazy", None)
        if lazy is not None:  # chosen on the device (gspx_graph_setup): downloaded when somebody asks
            self._perm_lazy = None
            if getattr(lazy, "_h", None):
                self._perm = lazy.download_perm()
            else:
                self._perm_done = False
        if not self._perm_done:
            self._perm_done = True
            mode = self.reorder
            if mode in (None, False, "none"):
                self._perm = None
            elif mode == "auto":
                # large graphs: Hilbert / Morton order of the coordinates if there are any, else reverse
                # Cuthill-McKee on the pattern (0.25 s at N = 1M; within 6 % of Morton on kNN
                # graphs) - kept only if it improves locality over the graph's own order
                big = self.n_vertices >= 4096
                self._perm = engine.auto_order(self.W, getattr(self, "coords", None), ctx=self.context) if big else None
            elif mode in ("morton", "hilbert"):
                self._perm = engine.locality_order(self.W, getattr(self, "coords", None), curve=mode,
                                                   ctx=self.context)
            elif mode == "rcm":
                self._perm = engine.locality_order(self.W, None)
            else:
                self._perm = np.asarray(mode, dtype=np.int32)
        return self._perm

    def device_graph(self, dtype=None):
        """The device-resident Laplacian in `dtype` (built on first use, cached)."""
        dt = np.dtype(dtype or self.compute_dtype)
        g = self._dev.get(dt)
        if g is None:
            perm = self._internal_order()
            if self.is_directed() and sparse.isspmatrix_csr(self.W):
                # (W + W.T) / 2 on the device, in the same call as the Laplacian (gspx_graph_setup)
                try:
                    g, _ = engine.DeviceGraph.setup(self.W, self.lap_type, dt, None, perm if perm is not None else "none",
                                                    ctx=self.context)
                except ValueError:
                    g = None
            if g is None:
                g = engine.DeviceGraph.from_w(self._symmetric_w(), self.lap_type, dtype=dt, perm=perm, ctx=self.context)
            if self.tiles == "auto":  # LDS tiles for the recurrence step, when the order is local
                self.tile_stats = g.auto_gather_tiles()
            elif self.tiles:
                self.tile_stats = g.enable_gather_tiles()
            self._dev[dt] = g
        return g

    def compute_laplacian(self, lap_type="combinatorial"):
        """graph.py:510-630.  The Laplacian is assembled by HIP kernels from W; ``G.L`` is the
        scipy CSR copy downloaded from the device on first access."""
        if lap_type not in ("combinatorial", "normalized"):
            raise ValueError("Unknown Laplacian type {}".format(lap_type))
        if lap_type != self.lap_type:
            self._forget_spectrum()
        self.lap_type = lap_type
        self._L = None
        self._internal_order()  # (an order chosen on the device is fetched before its graph goes)
        while self._dev:
            self._dev.popitem()[1].destroy()
        self.device_graph()  # build now: errors surface here, like in the reference

    @property
    def L(self):
        # the reference's L is float64 for normalized and for integer W, and follows W's dtype for
        # combinatorial; the device copy is in the compute dtype
        if self._L is None:
            self._L = self.device_graph().download_l()
        return self._L

    @property
    def dw(self):
        """Weighted degrees (graph.py:783-838), summed on the device while the Laplacian is built."""
        if self._dw is None:
            self._dw = self.device_graph().download_dw()
        return self._dw

    def _check_signal(self, s):
        """graph.py:632-640: anything array-like whose first axis runs over the vertices."""
        arr = np.asanyarray(s)
        if arr.ndim == 0 or arr.shape[0] != self.n_vertices:
            raise ValueError("First dimension must be the number of vertices "
                             "G.N = {}, got {}.".format(self.N, arr.shape))
        return arr

    def to_device(self, s, dtype=None):
        """Upload a signal ((N,), (N, Nsig) or (N, Nsig, Nfeat)) once and keep it on this graph's device: an
        engine.DeviceArray that Filter.filter / analyze / synthesize take and return, so that a chain of filters
        costs one upload and one download (np.asarray(result))."""
        arr = self._check_signal(s)
        return engine.DeviceArray.from_host(self.context, arr, np.dtype(dtype or self.compute_dtype))

    # ---- operators on the device CSR ----------------------------------------
    def dirichlet_energy(self, x):
        """x^T L x (graph.py:642-702), computed on the device: one sparse product and one reduction."""
        return self.device_graph().dirichlet_energy(self._check_signal(x))

    def get_edge_list(self):
        """(sources, targets, weights), graph.py:934-1029: all stored entries of a directed graph,
        the upper triangle (row-major) of an undirected one."""
        C = self.W.tocoo() if self.is_directed() else sparse.triu(self.W, format="coo")
        assert self.n_edges == C.row.size
        return C.row, C.col, C.data

    def _device_differential(self):
        """The device graph with the edge list grad / div / D work on.  An undirected graph without self-loops
        reads its edges off its own device Laplacian; a directed graph (all stored entries, D / sqrt(2),
        difference.py:160-161) or one with self-loops (diagonal entries are edges whose two D values cancel,
        difference.py:166) hands over Graph.get_edge_list, and the D values are formed on the device."""
        dev = self.device_graph()
        if (self.is_directed() or self.W.diagonal().any()) and getattr(self, "_edges_given_to", None) is not dev:
            sources, targets, weights = self.get_edge_list()
            dev.set_edge_list(sources, targets, weights, directed=self.is_directed())
            self._edges_given_to = dev
        return dev

    def compute_differential_operator(self):
        """difference.py:26-166.  D is assembled on the device (edge enumeration, sqrt of the
        weights) and cached there; ``G.D`` is the scipy csc copy."""
        self._D = self._device_differential().differential_operator()
        self._D_lap_type = self.lap_type

    @property
    def D(self):
        """Differential operator (difference.py:15-24), N x n_edges, L = D D^T."""
        stale = getattr(self, "_D", None) is None or getattr(self, "_D_lap_type", None) != self.lap_type
        if stale:
            self.logger.warning("The differential operator G.D is not available, we need to compute "
                                "it. Explicitly call G.compute_differential_operator() once "
                                "beforehand to suppress the warning.")
            self.compute_differential_operator()
        return self._D

    def grad(self, x):
        """Gradient D^T x of a vertex signal (difference.py:168-244), on the device."""
        return self._device_differential().grad(self._check_signal(x))

    def div(self, y):
        """Divergence D y of an edge signal (difference.py:246-331), on the device."""
        edge_signal = np.asanyarray(y)
        if edge_signal.shape[0] != self.Ne:
            raise ValueError("First dimension must be the number of edges "
                             "G.Ne = {}, got {}.".format(self.Ne, edge_signal.shape))
        return self._device_differential().div(edge_signal)

    # ---- spectrum bounds (read-only input of the hot path) ----------------------------------------
    @property
    def lmax(self):
        if self._lmax is None:
            self.logger.warning("The largest eigenvalue G.lmax is not available, we need to "
                                "estimate it. Explicitly call G.estimate_lmax() or "
                                "G.compute_fourier_basis() once beforehand to suppress the warning.")
            self.estimate_lmax()
        return self._lmax

    def _lmax_lanczos_device(self):
        # Lanczos on the device (the reference calls ARPACK eigsh(tol=5e-3) on the host, graph.py:911-917,
        # with a random start vector: 3.3 s at N = 1M and not reproducible).  Same contract: an estimate
        # from below, increased by 1 % (graph.py:920); non-convergence is a ValueError (engine).
        ritz, _ = self.device_graph().lanczos_lmax(max_iter=80, tol=5e-4)
        if not ritz <= self._get_upper_bound() * (1 + 1e-6) + 1e-12:  # graph.py:919 (an assert there)
            raise ValueError("The Lanczos estimate {} exceeds the upper bound of lambda_max.".format(ritz))
        return ritz * 1.01

    def _lmax_lanczos_host(self):
        try:
            ritz = splinalg.eigsh(self.L.astype(np.float64), k=1, tol=5e-3, ncv=min(self.N, 10),
                                  return_eigenvectors=False)[0]
        except splinalg.ArpackNoConvergence:
            raise ValueError("The Lanczos method did not converge. Try to use bounds.")
        assert ritz <= self._get_upper_bound() + 1e-12
        return ritz * 1.01  # 1 % safety margin, as the reference

    def estimate_lmax(self, method="lanczos"):
        """graph.py:858-931: 'lanczos' (device), 'lanczos-host' (ARPACK, the reference's own call) or
        'bounds'; a repeated call with the method already used is a no-op."""
        if method == self._lmax_method:
            return
        estimators = {"lanczos": self._lmax_lanczos_device, "lanczos-host": self._lmax_lanczos_host,
                      "bounds": self._get_upper_bound}
        if method not in estimators:
            raise ValueError("Unknown method {}".format(method))
        self._lmax_method = method
        self._lmax = estimators[method]()

    def _get_upper_bound(self):
        """The smallest of four classical upper bounds of lambda_max (graph.py:933-960)."""
        if self.lap_type != "combinatorial":
            return 2  # normalized Laplacian
        w_dtype = self._adj_dev.weights if self._adj_dev is not None else self.W.dtype
        if not self.is_directed() and w_dtype in (np.float64, np.int64):
            # the same four candidates from one device pass over W (gspx_graph_lmax_bounds), same operation order
            on_device = self.device_graph().lmax_bounds() if self.compute_dtype == np.float64 else None
            if on_device is not None:
                candidates = [self.n_vertices * on_device[0], 2 * on_device[1]]
                if self.n_edges:
                    candidates.append(on_device[2])
                candidates.append(on_device[3])  # (NaN with isolated vertices: never the minimum, as in the reference)
                return float(min(candidates))
        W, deg = self._symmetric_w(), np.asarray(self.dw, dtype=np.float64)
        # (the first candidate on W AS STORED, graph.py:941 - for a directed graph max W_ij, not max (W + W.T) / 2,
        # which would give a smaller bound than the reference's where this candidate is the minimum: ADVICE r5)
        candidates = [self.n_vertices * self.W.max(), 2 * deg.max()]
        if self.n_edges:
            ends = self.W.tocoo()  # max over edges: both triangles give the same maximum
            candidates.append((deg[ends.row] + deg[ends.col]).max())
        with np.errstate(divide="ignore", invalid="ignore"):
            candidates.append(np.max(deg + W.dot(deg) / deg))
        return min(candidates)

    def compute_fourier_basis(self):
        """Dense eigendecomposition (fourier.py:97-195) - host LAPACK, for small graphs / tests."""
        if self._U is None:
            lam, vec = np.linalg.eigh(self.L.toarray().astype(np.float64))
            if abs(lam[0]) < 1e-9:
                lam[0] = 0
            self._e, self._U = lam, vec
            self._lmax, self._lmax_method = lam[-1], "fourier"

    @property
    def e(self):
        self.compute_fourier_basis()
        return self._e

    @property
    def U(self):
        self.compute_fourier_basis()
        return self._U

    def __repr__(self):
        return "{}(n_vertices={}, n_edges={}, lap_type={})".format(
            type(self).__name__, self.n_vertices, self.n_edges, self.lap_type)


# ------------------------------------------------------------------------------------------------
# synthetic graph generators (vectorised restatements of the reference's models)
# ------------------------------------------------------------------------------------------------
def _sensor_points(N, seed, distributed=False):
    """Coordinates of ``graphs.Sensor`` (nngraphs/sensor.py:56-70), drawn from the same numpy stream as
    the reference so that the same seed gives the same graph: uniform in the unit square, or one jittered
    point per cell of a sqrt(N) x sqrt(N) grid."""
    stream = np.random.default_rng(seed)
    if not distributed:
        return stream.uniform(0, 1, (N, 2))
    side = np.sqrt(N)
    if side != np.floor(side):
        raise ValueError("The number of vertices must be a perfect square if they are to be "
                         "distributed on a grid.")
    cells = np.mgrid[0:1:1 / side, 0:1:1 / side].reshape(2, -1).T
    return cells + stream.uniform(0, 1 / side, (N, 2))


def sensor_weights(N, k=6, seed=None, return_coords=True):
    """W of ``graphs.Sensor(N, k, seed=seed)`` built on the HOST with a KD-tree (nngraphs/sensor.py:50-75,
    nngraph.py:213-297): k nearest neighbours of uniform points, weights exp(-d^2 / sigma) with sigma the
    mean neighbour distance, symmetrised by averaging.  The reference construction the device k-NN path is
    compared with."""
    coords = _sensor_points(N, seed)
    dist, nbr = spatial.cKDTree(coords).query(coords, k=k + 1, workers=-1)
    dist, nbr = dist[:, 1:], nbr[:, 1:]  # column 0 is the point itself
    sigma = float(dist.mean())
    directed = sparse.csc_matrix((np.exp(-np.square(dist.ravel()) / sigma),
                                  (np.repeat(np.arange(N), k), nbr.ravel())), shape=(N, N))
    W = sparse.csr_matrix((directed + directed.T) / 2)
    return (W, coords) if return_coords else W


_MINKOWSKI = {1: "manhattan", 2: "euclidean", np.inf: "max_dist"}


class NNGraph(Graph):
    """Nearest-neighbour graph from a point cloud (nngraphs/nngraph.py:13-313), built on the device
    (NNtype='knn', the KD-tree query, and NNtype='radius', the ball query: 1 to 64 dimensions), dist_type
    'euclidean' / 'manhattan' / 'max_dist', every symmetrize_type of utils.symmetrize.  Other settings
    raise NotImplementedError (no host fallback)."""

    def __init__(self, Xin, NNtype="knn", use_flann=False, center=True, rescale=True, k=10, sigma=None,
                 epsilon=0.01, plotting={}, symmetrize_type="average", dist_type="euclidean", order=0,
                 **kwargs):
        self.Xin = np.asanyarray(Xin)
        self.NNtype, self.use_flann, self.center, self.rescale = NNtype, use_flann, center, rescale
        self.k, self.sigma, self.epsilon = k, sigma, epsilon
        self.symmetrize_type, self.dist_type, self.order = symmetrize_type, dist_type, order
        n_points, dim = self.Xin.shape
        if n_points <= k:
            raise ValueError("The number of neighbors (k={}) must be smaller "
                             "than the number of nodes ({}).".format(k, n_points))
        if NNtype not in ("knn", "radius"):
            raise ValueError("Unknown NNtype {}".format(NNtype))
        # nngraph.py:139-145: 'minkowski' is the p-norm of the given order
        metric = _MINKOWSKI.get(order) if dist_type == "minkowski" else (
            dist_type if dist_type in engine.METRICS else None)
        if metric is None:
            raise NotImplementedError("the device builder covers dist_type 'euclidean', 'manhattan', 'max_dist' "
                                      "(and 'minkowski' of order 1, 2, inf)")
        if symmetrize_type not in engine.SYMMETRIZE:
            raise ValueError("Unknown symmetrization method {}.".format(symmetrize_type))  # utils.py:277
        points = self.Xin
        if center:  # nngraph.py:129-130: the cloud's centroid moves to the origin
            points = points - points.mean(axis=0)[np.newaxis, :]
        if rescale:  # nngraph.py:132-137: N^(1/min(d,3)) / 10 over half the bounding-box diagonal
            half_diagonal = 0.5 * np.linalg.norm(points.max(axis=0) - points.min(axis=0), 2)
            points = points * (np.power(n_points, 1.0 / float(min(dim, 3))) / 10.0 / half_diagonal)
        ctx = kwargs.get("ctx") or engine.default_context(int(kwargs.get("device", 0)))
        if NNtype == "knn":
            W, self.sigma, info = engine.knn_graph(points, k, sigma, ctx=ctx, metric=metric,
                                                   symmetrize=symmetrize_type, keep_on_device=True)
        else:  # nngraph.py:228-287; a symmetric relation: (W + W.T) / 2 = W
            W, self.sigma, info = engine.radius_graph(points, epsilon, sigma, ctx=ctx, metric=metric,
                                                      keep_on_device=True)
        self.knn_build_ms = info["build_ms"]
        Graph.__init__(self, W, plotting=plotting, coords=points, **kwargs)


class Sensor(NNGraph):
    """Random sensor graph (nngraphs/sensor.py:11-78): uniform points in the unit square, k nearest
    neighbours; the neighbour search runs on the device.  ``sensor_weights`` is the host (KD-tree)
    construction of the same matrix."""

    def __init__(self, N=64, k=6, distributed=False, seed=None, **kwargs):
        self.distributed, self.seed = distributed, seed
        NNGraph.__init__(self, Xin=_sensor_points(N, seed, distributed), k=k, rescale=False, center=False,
                         plotting={"limits": np.array([0, 1, 0, 1])}, **kwargs)


def _sample_pairs_within(rng, n, p):
    """m ~ Binomial(n(n-1)/2, p) distinct unordered pairs (r > c) among n vertices."""
    total = n * (n - 1) // 2
    m = rng.binomial(total, p) if total > 0 else 0
    if m == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    got = np.empty(0, dtype=np.int64)
    while got.size < m:
        cand = rng.integers(0, total, size=int((m - got.size) * 1.1) + 16)
        got = np.unique(np.concatenate([got, cand]))
    got = rng.permutation(got)[:m]
    # linear index -> (r, c) with r > c :  idx = r(r-1)/2 + c
    r = np.floor((1 + np.sqrt(1 + 8 * got.astype(np.float64))) / 2).astype(np.int64)
    r = np.where(r * (r - 1) // 2 > got, r - 1, r)
    r = np.where((r + 1) * r // 2 <= got, r + 1, r)
    c = got - r * (r - 1) // 2
    return r, c


def _sample_pairs_between(rng, na, nb, p):
    total = na * nb
    m = rng.binomial(total, p) if total > 0 else 0
    if m == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    got = np.empty(0, dtype=np.int64)
    while got.size < m:
        cand = rng.integers(0, total, size=int((m - got.size) * 1.1) + 16)
        got = np.unique(np.concatenate([got, cand]))
    got = rng.permutation(got)[:m]
    return got // nb, got % nb


def _block_probabilities(k, p, q):
    """The k x k matrix of edge probabilities from the within-block probabilities p (scalar or length-k
    vector) and the between-block ones q (scalar or k x k matrix; default 0.3 / k), as
    stochasticblockmodel.py:91-116 assembles it."""
    within = np.asanyarray(p, dtype=np.float64)
    if within.size == 1:
        within = np.full(k, within.reshape(-1)[0])
    if within.shape != (k,):
        raise ValueError("Optional parameter p is neither a scalar nor a vector of length k.")
    between = np.asanyarray(0.3 / k if q is None else q, dtype=np.float64)
    if between.size == 1:
        between = np.full((k, k), float(between.reshape(-1)[0]))
    if between.shape != (k, k):
        raise ValueError("Optional parameter q is neither a scalar nor a matrix of size k x k.")
    M = np.array(between, dtype=np.float64)
    np.fill_diagonal(M, within)
    return M


def sbm_weights(N, k=5, z=None, p=0.7, q=None, seed=None):
    """W with the distribution of ``graphs.StochasticBlockModel(N, k, z, p=p, q=q,
    directed=False, self_loops=False)`` (stochasticblockmodel.py:61-144): every unordered pair
    is an edge independently with probability p (same block) or q (different blocks); unit
    weights (int64, as the reference).  O(nnz) instead of the reference's O(N^2) loop.  Host / numpy
    sampler; the product's generators use the device sampler (engine.sbm_graph)."""
    stream = np.random.default_rng(seed)
    labels = np.sort(stream.integers(0, k, N)) if z is None else np.asarray(z)
    between = 0.3 / k if q is None else q
    by_block = np.argsort(labels, kind="stable")
    starts = np.searchsorted(labels[by_block], np.arange(k + 1))
    members = [by_block[starts[b]:starts[b + 1]] for b in range(k)]
    heads, tails = [], []
    for a in range(k):
        r, c = _sample_pairs_within(stream, members[a].size, p)
        heads.append(members[a][r])
        tails.append(members[a][c])
        for b in range(a):
            r, c = _sample_pairs_between(stream, members[a].size, members[b].size, between)
            heads.append(members[a][r])
            tails.append(members[b][c])
    heads = np.concatenate(heads) if heads else np.empty(0, np.int64)
    tails = np.concatenate(tails) if tails else np.empty(0, np.int64)
    W = sparse.csr_matrix((np.ones(2 * heads.size, dtype=np.int64),
                           (np.concatenate([heads, tails]), np.concatenate([tails, heads]))), shape=(N, N))
    return W, labels


class StochasticBlockModel(Graph):
    """stochasticblockmodel.py:12-181: labels z (sorted random labels from the same numpy stream as the
    reference when not given), probabilities M (or p on the diagonal and q elsewhere), edges sampled on the
    device in O(edges) (engine.sbm_graph) - equal in distribution to the reference's N^2 loop, not bit-equal.
    directed / self_loops / connected (n_try fresh samples until one is connected) as the reference.
    ``sbm_weights`` is the numpy sampler of the undirected, loop-free distribution."""

    def __init__(self, N=1024, k=5, z=None, M=None, p=0.7, q=None, directed=False, self_loops=False,
                 connected=False, n_try=10, seed=None, **kwargs):
        self.k, self.directed, self.self_loops, self.connected = k, directed, self_loops, connected
        self.n_try, self.seed = n_try, seed
        stream = np.random.default_rng(seed)
        # labels first, then the device sampler's seed: the reference draws z first too
        # (stochasticblockmodel.py:84-87), so a given seed yields the reference's blocks
        self.z = np.sort(stream.integers(0, k, N)) if z is None else np.asarray(z)
        if M is None:
            self.p, self.q = p, (0.3 / k if q is None else q)
            M = _block_probabilities(k, p, q)
        self.M = np.asarray(M, dtype=np.float64)
        if self.M.min() < 0 or self.M.max() > 1:
            raise ValueError("Probabilities should be in [0, 1].")
        ctx = kwargs.get("ctx") or engine.default_context(int(kwargs.get("device", 0)))
        tries, self.sampler_ms = n_try, 0.0
        while True:  # stochasticblockmodel.py:122-157: sample until connected, at most n_try times (None: for ever)
            device_seed = int(stream.integers(0, 2 ** 63))
            # (W stays on the device until somebody reads G.W: unit int64 weights then, like the reference's W)
            W, ms = engine.sbm_graph(self.z, self.M, seed=device_seed, ctx=ctx, keep_on_device=True,
                                     directed=directed, self_loops=self_loops)
            self.sampler_ms += ms
            if not connected:
                break
            trial = Graph(W, reorder="none", tiles=False, ctx=ctx)
            if trial.is_connected():
                W = trial.W  # (the host copy the connectivity test made: the device handle went with it)
                break
            if tries is not None:
                tries -= 1
                if tries <= 0:
                    raise ValueError("The graph could not be connected after {} trials. Increase the connection "
                                     "probability or the number of trials.".format(self.n_try))
        self.info = {"node_com": self.z, "comm_sizes": np.bincount(self.z), "world_rad": np.sqrt(N)}
        Graph.__init__(self, W, **kwargs)


class ErdosRenyi(StochasticBlockModel):
    """erdosrenyi.py:49-61: the one-block stochastic block model."""

    def __init__(self, N=100, p=0.1, directed=False, self_loops=False, connected=False, n_try=10,
                 seed=None, **kwargs):
        StochasticBlockModel.__init__(self, N=N, k=1, p=p, directed=directed, self_loops=self_loops,
                                      connected=connected, n_try=n_try, seed=seed, **kwargs)
