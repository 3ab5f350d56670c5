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


class Graph:
    def __init__(self, adjacency, lap_type="combinatorial", coords=None, plotting={}, *,
                 compute_dtype=np.float64, device=0, reorder="auto", tiles="auto"):
        self.logger = _logger
        if not sparse.issparse(adjacency):
            adjacency = np.asanyarray(adjacency)
        if adjacency.ndim != 2 or adjacency.shape[0] != adjacency.shape[1]:
            raise ValueError("Adjacency: must be a square matrix.")
        self._adjacency = sparse.csr_matrix(adjacency, copy=False)
        total = self._adjacency.sum()
        if np.isnan(total):
            raise ValueError("Adjacency: there is a Not a Number (NaN).")
        if np.isinf(total):
            raise ValueError("Adjacency: there is an infinite value.")
        if self._adjacency.diagonal().any():
            self.logger.warning("Adjacency: there are self-loops (non-zeros on the diagonal). "
                                "The Laplacian will not see them.")
        if (self._adjacency < 0).nnz != 0:
            self.logger.warning("Adjacency: there are negative edge weights.")
        self.n_vertices = self._adjacency.shape[0]
        self._adjacency.eliminate_zeros()

        self._directed = None
        if self.is_directed():
            self.n_edges = self._adjacency.nnz
        else:
            diagonal = int(np.count_nonzero(self._adjacency.diagonal()))
            self.n_edges = (self._adjacency.nnz - diagonal) // 2 + diagonal
        if coords is not None:
            self.coords = np.asanyarray(coords)
        self.plotting = dict(plotting)
        self.signals = dict()

        self.compute_dtype = np.dtype(compute_dtype)
        self.device = int(device)
        self.reorder = reorder
        self.tiles = tiles  # "auto" | True | False: gather tiles of the LDS-staged recurrence step
        self.tile_stats = None
        self._perm = None
        self._perm_done = False

        self._dw = None
        self._lmax = None
        self._lmax_method = None
        self._U = None
        self._e = None
        self._L = None
        self._dev = {}

        self.lap_type = lap_type
        self.compute_laplacian(lap_type)
        self.Ne = self.n_edges
        self.N = self.n_vertices

    # ---- adjacency -----------------------------------------------------------------------------
    @property
    def W(self):
        return self._adjacency

    def is_directed(self):
        if self._directed is None:
            self._directed = (self.W != self.W.T).nnz != 0
        return self._directed

    def _symmetric_w(self):
        """W itself if undirected, else (W + W.T)/2 (utils.symmetrize 'average', graph.py:613-616)."""
        if not self.is_directed():
            return self.W
        return sparse.csr_matrix((self.W + self.W.T) / 2)

    # ---- Laplacian (device) --------------------------------------------------------------------
    def _internal_order(self):
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
                self._perm = engine.auto_order(self.W, getattr(self, "coords", None), self.device) if big else None
            elif mode in ("morton", "hilbert"):
                self._perm = engine.locality_order(self.W, getattr(self, "coords", None), curve=mode,
                                                   device=self.device)
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
            ctx = engine.default_context(self.device)
            g = engine.DeviceGraph.from_w(self._symmetric_w(), self.lap_type, dtype=dt,
                                          perm=self._internal_order(), ctx=ctx)
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
            # caches invalidated when the Laplacian changes (graph.py:602-609)
            self._lmax = None
            self._lmax_method = None
            self._U = None
            self._e = None
        self.lap_type = lap_type
        self._L = None
        for g in self._dev.values():
            g.destroy()
        self._dev = {}
        self.device_graph()  # build now: errors surface here, like in the reference

    @property
    def L(self):
        if self._L is None:
            # the reference's L is float64 for normalized and for integer W, and follows W's
            # dtype for combinatorial; the device copy is in the compute dtype
            self._L = self.device_graph().download_l()
        return self._L

    @property
    def dw(self):
        if self._dw is None:
            self._dw = self.device_graph().download_dw()
        return self._dw

    def _check_signal(self, s):
        s = np.asanyarray(s)
        if s.shape[0] != self.n_vertices:
            raise ValueError("First dimension must be the number of vertices "
                             "G.N = {}, got {}.".format(self.N, s.shape))
        return s

    # ---- operators on the device CSR ----------------------------------------
    def dirichlet_energy(self, x):
        """x^T L x (graph.py:642-702), computed on the device: one sparse product and one reduction."""
        x = self._check_signal(x)
        return self.device_graph().dirichlet_energy(x)

    def get_edge_list(self):
        """(sources, targets, weights), graph.py:934-1029: all stored entries of a directed graph,
        the upper triangle (row-major) of an undirected one."""
        if self.is_directed():
            C = self.W.tocoo()
        else:
            C = sparse.triu(self.W, format="coo")
        assert self.n_edges == C.row.size
        return C.row, C.col, C.data

    def _device_differential(self):
        if self.is_directed():
            raise NotImplementedError("the device differential operator covers undirected graphs; "
                                      "this graph is directed")
        if self.W.diagonal().any():
            raise NotImplementedError("the device differential operator does not cover self-loops")
        return self.device_graph()

    def compute_differential_operator(self):
        """difference.py:26-166.  D is assembled on the device (edge enumeration, sqrt of the
        weights) and cached there; ``G.D`` is the scipy csc copy."""
        dev = self._device_differential()
        self._D = dev.differential_operator()
        self._D_lap_type = self.lap_type

    @property
    def D(self):
        """Differential operator (difference.py:15-24), N x n_edges, L = D D^T."""
        if getattr(self, "_D", None) is None or getattr(self, "_D_lap_type", None) != self.lap_type:
            self.logger.warning("The differential operator G.D is not available, we need to compute "
                                "it. Explicitly call G.compute_differential_operator() once "
                                "beforehand to suppress the warning.")
            self.compute_differential_operator()
        return self._D

    def grad(self, x):
        """Gradient D^T x of a vertex signal (difference.py:168-244), on the device."""
        x = self._check_signal(x)
        return self._device_differential().grad(x)

    def div(self, y):
        """Divergence D y of an edge signal (difference.py:246-331), on the device."""
        y = np.asanyarray(y)
        if y.shape[0] != self.Ne:
            raise ValueError("First dimension must be the number of edges "
                             "G.Ne = {}, got {}.".format(self.Ne, y.shape))
        return self._device_differential().div(y)

    # ---- spectrum bounds (host; read-only input of the hot path) -------------------------------
    @property
    def lmax(self):
        if self._lmax is None:
            self.logger.warning("The largest eigenvalue G.lmax is not available, we need to "
                                "estimate it. Explicitly call G.estimate_lmax() or "
                                "G.compute_fourier_basis() once beforehand to suppress the warning.")
            self.estimate_lmax()
        return self._lmax

    def estimate_lmax(self, method="lanczos"):
        if method == self._lmax_method:
            return
        self._lmax_method = method
        if method == "lanczos":
            # Lanczos on the device (the reference calls ARPACK eigsh(tol=5e-3) on the host,
            # graph.py:911-917, with a random start vector: 3.3 s at N = 1M and not reproducible).
            # Same contract: an estimate from below, increased by 1 % (graph.py:920).
            lmax, _ = self.device_graph().lanczos_lmax(max_iter=80, tol=5e-4)
            assert lmax <= self._get_upper_bound() * (1 + 1e-6) + 1e-12
            self._lmax = lmax * 1.01
        elif method == "lanczos-host":
            try:
                lmax = splinalg.eigsh(self.L.astype(np.float64), k=1, tol=5e-3,
                                      ncv=min(self.N, 10), return_eigenvectors=False)[0]
            except splinalg.ArpackNoConvergence:
                raise ValueError("The Lanczos method did not converge. Try to use bounds.")
            assert lmax <= self._get_upper_bound() + 1e-12
            self._lmax = lmax * 1.01  # 1 % safety margin, as the reference
        elif method == "bounds":
            self._lmax = self._get_upper_bound()
        else:
            raise ValueError("Unknown method {}".format(method))

    def _get_upper_bound(self):
        if self.lap_type == "normalized":
            return 2
        W = self._symmetric_w()
        dw = np.asarray(self.dw, dtype=np.float64)
        bounds = [self.n_vertices * W.max(), 2 * dw.max()]
        if self.n_edges > 0:
            coo = self.W.tocoo()  # max over edges: both triangles give the same maximum
            bounds.append(np.max(dw[coo.row] + dw[coo.col]))
        with np.errstate(divide="ignore", invalid="ignore"):
            m = W.dot(dw) / dw
        bounds.append(np.max(dw + m))
        return min(bounds)

    def compute_fourier_basis(self):
        """Dense eigendecomposition (fourier.py:97-195) - host LAPACK, for small graphs / tests."""
        if self._U is not None:
            return
        e, U = np.linalg.eigh(self.L.toarray().astype(np.float64))
        e[0] = 0 if abs(e[0]) < 1e-9 else e[0]
        self._e, self._U = e, U
        self._lmax = e[-1]
        self._lmax_method = "fourier"

    @property
    def e(self):
        if self._e is None:
            self.compute_fourier_basis()
        return self._e

    @property
    def U(self):
        if self._U is None:
            self.compute_fourier_basis()
        return self._U

    def __repr__(self):
        return "{}(n_vertices={}, n_edges={}, lap_type={})".format(
            self.__class__.__name__, self.n_vertices, self.n_edges, self.lap_type)


# ------------------------------------------------------------------------------------------------
# synthetic graph generators (vectorised restatements of the reference's models)
# ------------------------------------------------------------------------------------------------
def sensor_weights(N, k=6, seed=None, return_coords=True):
    """W of ``graphs.Sensor(N, k, seed=seed)`` (nngraphs/sensor.py:50-75, nngraph.py:213-297):
    uniform coordinates in the unit square, k nearest neighbours, weights exp(-d^2/sigma) with
    sigma = mean neighbour distance, symmetrised by averaging."""
    rng = np.random.default_rng(seed)
    coords = rng.uniform(0, 1, (N, 2))
    tree = spatial.cKDTree(coords)
    D, NN = tree.query(coords, k=k + 1, workers=-1)
    sigma = np.mean(D[:, 1:])
    rows = np.repeat(np.arange(N), k)
    cols = NN[:, 1:].ravel()
    vals = np.exp(-np.power(D[:, 1:].ravel(), 2) / float(sigma))
    W = sparse.csc_matrix((vals, (rows, cols)), shape=(N, N))
    W = sparse.csr_matrix((W + W.T) / 2)
    return (W, coords) if return_coords else W


class NNGraph(Graph):
    """Nearest-neighbour graph from a point cloud (nngraphs/nngraph.py:13-313), built on the device
    in 1-3 dimensions: NNtype='knn' (KD-tree query) and NNtype='radius' (ball query), dist_type
    'euclidean' / 'manhattan' / 'max_dist', every symmetrize_type of utils.symmetrize.  Other settings
    raise NotImplementedError (no host fallback)."""

    def __init__(self, Xin, NNtype="knn", use_flann=False, center=True, rescale=True, k=10, sigma=None,
                 epsilon=0.01, plotting={}, symmetrize_type="average", dist_type="euclidean", order=0,
                 **kwargs):
        self.Xin = np.asanyarray(Xin)
        self.NNtype, self.use_flann, self.center, self.rescale = NNtype, use_flann, center, rescale
        self.k, self.sigma, self.epsilon = k, sigma, epsilon
        self.symmetrize_type, self.dist_type, self.order = symmetrize_type, dist_type, order
        N, d = np.shape(self.Xin)
        Xout = self.Xin
        if k >= N:
            raise ValueError("The number of neighbors (k={}) must be smaller "
                             "than the number of nodes ({}).".format(k, N))
        if NNtype not in ("knn", "radius"):
            raise ValueError("Unknown NNtype {}".format(NNtype))
        # nngraph.py:139-145: 'minkowski' is the p-norm of the given order
        metric = {"euclidean": "euclidean", "manhattan": "manhattan", "max_dist": "max_dist"}.get(dist_type)
        if dist_type == "minkowski":
            metric = {1: "manhattan", 2: "euclidean", np.inf: "max_dist"}.get(order)
        if metric is None:
            raise NotImplementedError("the device builder covers dist_type 'euclidean', 'manhattan', 'max_dist' "
                                      "(and 'minkowski' of order 1, 2, inf)")
        if symmetrize_type not in engine.SYMMETRIZE:
            raise ValueError("Unknown symmetrization method {}.".format(symmetrize_type))  # utils.py:277
        if self.center:  # nngraph.py:129-130
            Xout = self.Xin - np.kron(np.ones((N, 1)), np.mean(self.Xin, axis=0))
        if self.rescale:  # nngraph.py:132-137
            bounding_radius = 0.5 * np.linalg.norm(np.amax(Xout, axis=0) - np.amin(Xout, axis=0), 2)
            scale = np.power(N, 1.0 / float(min(d, 3))) / 10.0
            Xout = Xout * (scale / bounding_radius)
        ctx = engine.default_context(int(kwargs.get("device", 0)))
        if NNtype == "knn":
            W, self.sigma, info = engine.knn_graph(Xout, k, sigma, ctx=ctx, metric=metric,
                                                   symmetrize=symmetrize_type)
        else:  # nngraph.py:228-287; a symmetric relation: (W + W.T) / 2 = W
            W, self.sigma, info = engine.radius_graph(Xout, epsilon, sigma, ctx=ctx, metric=metric)
        self.knn_build_ms = info["build_ms"]
        super().__init__(W, plotting=plotting, coords=Xout, **kwargs)


class Sensor(NNGraph):
    """Random sensor graph (nngraphs/sensor.py:11-78): uniform points in the unit square, k nearest
    neighbours; the neighbour search runs on the device.  ``sensor_weights`` is the host (KD-tree)
    construction of the same matrix."""

    def __init__(self, N=64, k=6, distributed=False, seed=None, **kwargs):
        self.distributed = distributed
        self.seed = seed
        rng = np.random.default_rng(self.seed)
        if distributed:
            m = np.sqrt(N)
            if not m.is_integer():
                raise ValueError("The number of vertices must be a perfect square if they are to be "
                                 "distributed on a grid.")
            coords = np.mgrid[0:1:1 / m, 0:1:1 / m].reshape(2, -1).T
            coords += rng.uniform(0, 1 / m, (N, 2))
        else:
            coords = rng.uniform(0, 1, (N, 2))
        plotting = {"limits": np.array([0, 1, 0, 1])}
        super().__init__(Xin=coords, k=k, rescale=False, center=False, plotting=plotting, **kwargs)


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


def sbm_weights(N, k=5, z=None, p=0.7, q=None, seed=None):
    """W with the distribution of ``graphs.StochasticBlockModel(N, k, z, p=p, q=q,
    directed=False, self_loops=False)`` (stochasticblockmodel.py:61-144): every unordered pair
    is an edge independently with probability p (same block) or q (different blocks); unit
    weights (int64, as the reference).  O(nnz) instead of the reference's O(N^2) loop."""
    rng = np.random.default_rng(seed)
    if z is None:
        z = np.sort(rng.integers(0, k, N))
    z = np.asarray(z)
    if q is None:
        q = 0.3 / k
    order = np.argsort(z, kind="stable")
    bounds = np.searchsorted(z[order], np.arange(k + 1))
    rows, cols = [], []
    for a in range(k):
        ia = order[bounds[a]:bounds[a + 1]]
        r, c = _sample_pairs_within(rng, ia.size, p)
        rows.append(ia[r]); cols.append(ia[c])
        for b in range(a):
            ib = order[bounds[b]:bounds[b + 1]]
            r, c = _sample_pairs_between(rng, ia.size, ib.size, q)
            rows.append(ia[r]); cols.append(ib[c])
    rows = np.concatenate(rows) if rows else np.empty(0, np.int64)
    cols = np.concatenate(cols) if cols else np.empty(0, np.int64)
    data = np.ones(rows.size * 2, dtype=np.int64)
    W = sparse.csr_matrix((data, (np.concatenate([rows, cols]), np.concatenate([cols, rows]))),
                          shape=(N, N))
    return W, z


class StochasticBlockModel(Graph):
    """stochasticblockmodel.py:12-181 for undirected graphs without self-loops: labels z (sorted random
    labels from the same numpy stream as the reference when not given), probabilities M (or p on the
    diagonal and q elsewhere), edges sampled on the device in O(edges) (engine.sbm_graph) - equal in
    distribution to the reference's N^2 loop, not bit-equal.  ``sbm_weights`` is the numpy sampler of
    the same distribution."""

    def __init__(self, N=1024, k=5, z=None, M=None, p=0.7, q=None, directed=False, self_loops=False,
                 connected=False, n_try=10, seed=None, **kwargs):
        if directed or self_loops or connected:
            raise NotImplementedError("the device sampler covers directed=False, self_loops=False, "
                                      "connected=False")
        self.k, self.directed, self.self_loops, self.connected = k, directed, self_loops, connected
        self.n_try, self.seed = n_try, seed
        rng = np.random.default_rng(seed)
        if z is None:
            z = rng.integers(0, k, N)
            z.sort()  # stochasticblockmodel.py:84-87
        self.z = np.asarray(z)
        if M is None:
            self.p = p
            p = np.asanyarray(p, dtype=np.float64)
            if p.size == 1:
                p = p * np.ones(k)
            if p.shape != (k,):
                raise ValueError("Optional parameter p is neither a scalar nor a vector of length k.")
            if q is None:
                q = 0.3 / k
            self.q = q
            q = np.asanyarray(q, dtype=np.float64)
            if q.size == 1:
                q = q * np.ones((k, k))
            if q.shape != (k, k):
                raise ValueError("Optional parameter q is neither a scalar nor a matrix of size k x k.")
            M = np.array(q, dtype=np.float64)
            M.flat[::k + 1] = p
        self.M = np.asarray(M, dtype=np.float64)
        if (self.M < 0).any() or (self.M > 1).any():
            raise ValueError("Probabilities should be in [0, 1].")
        sub = int(rng.integers(0, 2 ** 63))  # the device stream's seed, drawn from the same generator
        ctx = engine.default_context(int(kwargs.get("device", 0)))
        W, self.sampler_ms = engine.sbm_graph(self.z, self.M, seed=sub, ctx=ctx)
        W = sparse.csr_matrix((np.ones(W.nnz, dtype=np.int64), W.indices, W.indptr), shape=W.shape)  # int64 unit
        super().__init__(W, **kwargs)                                  # weights, as the reference's W


class ErdosRenyi(StochasticBlockModel):
    """erdosrenyi.py:49-61: the one-block stochastic block model."""

    def __init__(self, N=100, p=0.1, directed=False, self_loops=False, connected=False, n_try=10,
                 seed=None, **kwargs):
        super().__init__(N=N, k=1, p=p, directed=directed, self_loops=self_loops, connected=connected,
                         n_try=n_try, seed=seed, **kwargs)
