"""Parity of the operators next to the Chebyshev path through the C-ABI:
L x, Dirichlet energy, differential operator / grad / div, Tikhonov regression by CG.
Against the reference's golden vectors (tests/golden/ops_sensor123.npz), its doctest values and the
oracle (oracle/ops_oracle.py) on larger seeded graphs.  Needs a real MI355X: `-m gpu`.

Tolerances: float64 1e-5 relative, float32 1e-3 (BASELINE.json); the direct operators are asserted
far inside that.  The CG solutions are compared at 1e-6 / 2e-3: the solver itself stops at a 1e-5
relative residual, device and scipy follow the same recurrence.
"""
import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err
from oracle import ops_oracle as ops
from pygsp_amd import engine, graphs, learning
from gpu_helpers import random_graph

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.float32): 3e-5}
CGTOL = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 2e-3}


@pytest.fixture(scope="module")
def ctx():
    return engine.default_context(0)


def test_doctest_values():
    """graph.py:686-699, difference.py:104-118."""
    W = sparse.diags([np.ones(4), np.ones(4)], [1, -1]).tocsr()
    G = graphs.Graph(W)
    x = [0, 2, 2, 4, 4]
    assert G.dirichlet_energy(x) == 8.0
    G.compute_differential_operator()
    np.testing.assert_allclose(G.grad(x), [2, 0, 2, 0], atol=1e-15)
    G3 = graphs.Graph([[0, 2, 0], [2, 0, 1], [0, 1, 0]])
    G3.compute_differential_operator()
    np.testing.assert_allclose(G3.D.toarray(), [[-1.41421356, 0], [1.41421356, -1], [0, 1]], atol=1e-8)
    G3.compute_laplacian("normalized")
    G3.compute_differential_operator()
    np.testing.assert_allclose(G3.D.toarray(), [[-1, 0], [0.81649658, -0.57735027], [0, 1]], atol=1e-8)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_sensor123(golden_ops, dtype):
    g = golden_ops
    tol = TOL[np.dtype(dtype)]
    W = csr_from(g, "W")
    x, X5 = g["x"], g["X5"]
    for lt in ("combinatorial", "normalized"):
        G = graphs.Graph(W, lap_type=lt, compute_dtype=dtype)
        dev = G.device_graph()
        assert rel_err(dev.laplacian_apply(X5), g["Lx_" + lt]) < tol
        assert rel_err(dev.laplacian_apply(x), ops.laplacian(W, lt).dot(x)) < tol
        e = G.dirichlet_energy(x)
        assert abs(e - float(g["energy_" + lt])) <= tol * 10 * abs(float(g["energy_" + lt]))
        assert rel_err(G.dirichlet_energy(X5), g["energy5_" + lt]) < tol * 10
        src, dst, w = dev.edge_list()
        np.testing.assert_array_equal(src, g["edges_src"])
        np.testing.assert_array_equal(dst, g["edges_dst"])
        assert rel_err(w, g["edges_w"]) < tol
        assert G.Ne == src.size == dev.n_edges()
        G.compute_differential_operator()
        assert rel_err(G.D.toarray(), g["D_" + lt]) < tol
        assert rel_err(G.grad(x), g["grad_" + lt]) < tol
        assert rel_err(G.grad(X5), g["grad5_" + lt]) < tol
        assert rel_err(G.div(g["grad_" + lt]), g["div_" + lt]) < tol
        assert rel_err(G.div(g["grad5_" + lt]), g["div5_" + lt]) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_tikhonov(golden_ops, dtype):
    g = golden_ops
    tol = CGTOL[np.dtype(dtype)]
    G = graphs.Graph(csr_from(g, "W"), compute_dtype=dtype)
    mask = g["mask"]
    m0 = np.nan_to_num(g["measures"])
    for tau in (0.5, 5.0):
        assert rel_err(learning.regression_tikhonov(G, m0, mask, tau=tau), g["reg_tau%g" % tau]) < tol
    assert rel_err(learning.regression_tikhonov(G, g["reg3_in"], mask, tau=0.5), g["reg3_tau0.5"]) < tol
    lab = g["labels"].astype(float)
    lab[~mask] = np.nan
    rec = learning.classification_tikhonov(G, lab, mask, tau=0.1)
    assert rel_err(rec, g["class_tau0.1"]) < tol
    np.testing.assert_array_equal(np.argmax(rec, axis=1), np.argmax(g["class_tau0.1"], axis=1))
    with pytest.raises(NotImplementedError):
        learning.regression_tikhonov(G, m0, mask, tau=0)
    with pytest.raises(ValueError):
        learning.regression_tikhonov(G, m0, mask[:-1], tau=1.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("lap_type", ["combinatorial", "normalized"])
def test_oracle_sensor20k(ctx, dtype, lap_type):
    tol = TOL[np.dtype(dtype)] * 10
    rng = np.random.default_rng(5)
    N = 20000
    W, coords = graphs.sensor_weights(N, k=8, seed=3)
    L = ops.laplacian(W, lap_type)
    D = ops.differential_operator(W, lap_type)
    for perm in (engine.locality_order(W, coords), None):
        dev = engine.DeviceGraph.from_w(W, lap_type, dtype=dtype, perm=perm, ctx=ctx)
        if perm is not None:  # with gather tiles the products run on the LDS-staged kernel
            assert dev.build_gather_tiles()["slow_blocks"] == 0
        for nsig in (1, 3, 16, 64, 200):
            X = rng.standard_normal((N, nsig)).astype(dtype)
            X64 = X.astype(np.float64)
            assert rel_err(dev.laplacian_apply(X), L.dot(X64)) < tol, (nsig, "Lx")
            gram = dev.dirichlet_energy(X)
            assert rel_err(gram, ops.dirichlet_energy(L, X64)) < tol * 10, (nsig, "energy")
            Y = dev.grad(X)
            assert Y.shape == (D.shape[1], nsig)
            assert rel_err(Y, ops.grad(D, X64)) < tol, (nsig, "grad")
            Z = dev.div(Y)
            assert rel_err(Z, ops.div(D, Y.astype(np.float64))) < tol, (nsig, "div")
            ctx.set_option("edge_vertex_walk", 0)  # the edge-order kernels: same sums, same order
            try:
                assert rel_err(dev.grad(X), Y) < 4 * np.finfo(dtype).eps and rel_err(dev.div(Y), Z) < 4 * np.finfo(dtype).eps
            finally:
                ctx.set_option("edge_vertex_walk", 1)
        # div(grad(x)) = L x (difference.py:38-45)
        x = rng.standard_normal(N)
        assert rel_err(dev.div(dev.grad(x)), L.dot(x.astype(dtype).astype(np.float64))) < tol * 10
        assert rel_err(dev.differential_operator().toarray()[:50], D.toarray()[:50]) < tol
        dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_tikhonov(ctx, dtype):
    tol = CGTOL[np.dtype(dtype)]
    rng = np.random.default_rng(11)
    N = 20000
    W, coords = graphs.sensor_weights(N, k=8, seed=4)
    L = ops.laplacian(W)
    mask = rng.uniform(size=N) > 0.6
    smooth = np.sin(4 * coords[:, :1]) * np.cos(3 * coords[:, 1:]) + 0.1 * rng.standard_normal((N, 5))
    y = np.where(mask[:, None], smooth, 0.0)
    for perm in (engine.locality_order(W, coords), None):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        if perm is not None:
            dev.build_gather_tiles()
        for tau in (0.3, 3.0):
            ref = ops.regression_tikhonov(L, y.astype(dtype).astype(np.float64), mask, tau)
            x, iters, _ = dev.tikhonov_cg(tau, mask, y)
            assert x.shape == ref.shape and iters.shape == (5,)
            assert iters.min() >= 5 and iters.max() < 2000
            assert rel_err(x, ref) < tol, tau
            # the residual the solver promises: ||b - A x|| <= rtol ||b|| (+ rounding)
            A = sparse.diags(mask.astype(float)) + tau * L
            r = y - A.dot(x.astype(np.float64))
            bound = 1e-5 if dtype == np.float64 else 5e-5
            assert (np.linalg.norm(r, axis=0) <= bound * 1.01 * np.linalg.norm(y, axis=0)).all()
        # one column, and a zero right-hand side (0 iterations, x = 0)
        x1, it1, _ = dev.tikhonov_cg(1.0, mask, y[:, 0])
        assert x1.shape == (N,) and rel_err(x1, ops.regression_tikhonov(L, y[:, 0].copy(), mask, 1.0)) < tol
        x0, it0, _ = dev.tikhonov_cg(1.0, mask, np.zeros((N, 2)))
        assert not x0.any() and not it0.any()
        dev.destroy()


def test_ragged_graph_and_errors(ctx):
    Wr = random_graph(5000, 7, seed=41, hub=True, isolated=5)
    rng = np.random.default_rng(2)
    for lt in ("combinatorial", "normalized"):
        L = ops.laplacian(Wr, lt)
        D = ops.differential_operator(Wr, lt)
        G = graphs.Graph(Wr, lap_type=lt)
        X = rng.standard_normal((5000, 7))
        assert rel_err(G.device_graph().laplacian_apply(X), L.dot(X)) < 1e-12
        assert rel_err(G.dirichlet_energy(X), ops.dirichlet_energy(L, X)) < 1e-11
        assert rel_err(G.grad(X), ops.grad(D, X)) < 1e-12
        Y = rng.standard_normal((D.shape[1], 3))
        assert rel_err(G.div(Y), ops.div(D, Y)) < 1e-12
        with pytest.raises(ValueError):
            G.div(Y[:-1])
        with pytest.raises(ValueError):
            G.dirichlet_energy(X[:-1])
    Gd = graphs.Graph(sparse.csr_matrix(np.array([[0, 2, 0], [0, 0, 1], [0, 0, 0.0]])))
    assert abs(Gd.dirichlet_energy([0.0, 1.0, 3.0]) - ops.dirichlet_energy(ops.laplacian(Gd.W), np.array([0, 1, 3.0]))) < 1e-14
    # a graph uploaded as a Laplacian carries no degrees: no differential operator
    dev = engine.DeviceGraph.from_l(ops.laplacian(Wr), ctx=ctx)
    with pytest.raises(ValueError):
        dev.n_edges()
    assert np.max(np.abs(dev.laplacian_apply(np.ones(5000)))) < 1e-12  # L 1 = 0
    dev.destroy()


def test_doctest_values_directed():
    """difference.py:120-138: the directed 3-vertex example, both Laplacian types."""
    G = graphs.Graph([[0, 2, 0], [2, 0, 1], [0, 0, 0]])
    assert G.is_directed() and G.Ne == 3
    G.compute_differential_operator()
    np.testing.assert_allclose(G.D.toarray(), [[-1, 1, 0], [1, -1, -0.70710678], [0, 0, 0.70710678]], atol=1e-8)
    G.compute_laplacian("normalized")
    G.compute_differential_operator()
    np.testing.assert_allclose(G.D.toarray(), [[-0.70710678, 0.70710678, 0], [0.63245553, -0.63245553, -0.4472136],
                                               [0, 0, 1]], atol=1e-7)
    # L = D D^T also for directed graphs (difference.py:31)
    assert abs(G.D.dot(G.D.T) - G.L).max() < 1e-14


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_directed_and_self_loops(dtype):
    """difference.py:160-166 against the real reference (tests/golden/ops_directed.npz): D, grad, div of a directed
    weighted graph (every stored entry an edge, values / sqrt(2)) and of an undirected graph with self-loops (their
    stored zeros eliminated) - the edge list handed to the device, the D values formed there from its degrees."""
    from conftest import load_golden
    g = load_golden("ops_directed.npz")
    tol = TOL[np.dtype(dtype)]
    x, X4 = g["x"], g["X4"]
    for name in ("dir", "loops"):
        W = csr_from(g, "W" + name)
        for lt in ("combinatorial", "normalized"):
            key = "{}_{}".format(name, lt)
            G = graphs.Graph(W, lap_type=lt, compute_dtype=dtype)
            assert G.is_directed() == (name == "dir") and G.Ne == int(g["ne_" + name])
            assert rel_err(G.dw, g["dw_" + name]) < tol and rel_err(G.L.toarray(), g["L_" + key]) < tol
            G.compute_differential_operator()
            assert G.D.shape == g["D_" + key].shape and rel_err(G.D.toarray(), g["D_" + key]) < tol
            assert G.D.nnz == np.count_nonzero(g["D_" + key])  # the self-loops' zeros are not stored
            assert rel_err(G.grad(x), g["grad_" + key]) < tol and rel_err(G.grad(X4), g["grad4_" + key]) < tol
            assert rel_err(G.div(g["grad_" + key]), g["div_" + key]) < tol
            assert rel_err(G.div(g["grad4_" + key]), g["div4_" + key]) < tol
            src, dst, w = G.device_graph().edge_list()
            np.testing.assert_array_equal(src, g["src_" + name])
            np.testing.assert_array_equal(dst, g["dst_" + name])
            with pytest.raises(ValueError):
                G.div(np.ones(G.Ne + 1))
        # switching the Laplacian type rebuilds the device graph: the edge list follows it
        G.compute_laplacian("combinatorial")
        assert rel_err(G.grad(x), g["grad_{}_combinatorial".format(name)]) < tol


def test_directed_graph_at_size_against_the_oracle(ctx):
    """A directed random graph of 30k vertices (ragged rows, a hub), 6 signals: grad / div against the oracle's D;
    edge lists that are not in get_edge_list order are refused."""
    rng = np.random.default_rng(9)
    A = sparse.random(30000, 30000, 2e-4, random_state=4, format="csr")
    A.setdiag(0)
    A.eliminate_zeros()
    G = graphs.Graph(A)
    assert G.is_directed()
    D = ops.differential_operator(A)
    X = rng.standard_normal((G.N, 6))
    assert rel_err(G.grad(X), ops.grad(D, X)) < 1e-12
    Y = rng.standard_normal((G.Ne, 6))
    assert rel_err(G.div(Y), ops.div(D, Y)) < 1e-12
    src, dst, w = G.get_edge_list()
    with pytest.raises(ValueError):
        G.device_graph().set_edge_list(src[::-1], dst[::-1], w[::-1], True)
    with pytest.raises(ValueError):
        G.device_graph().set_edge_list(src, dst + G.N, w, True)
