"""SURVEY section 8 rows a1, a6-a9 against the REFERENCE'S OWN golden vectors (tests/golden/*.npz, generated
by importing the real pygsp: tests/golden/gen_golden.py), through the C-ABI on a real MI355X (`-m gpu`).
Runs first: these are the deterministic reference-fixture tests the coverage table is judged on.

  a1  Graph.compute_laplacian (graph.py:510-630): 4x4 fixtures of pygsp/tests/test_graphs.py:195-254,
      Logo and Sensor(123) bit for bit, empty / self-loop-only graphs (test_graphs.py:432-461)
  a6  cheby_op (approximations.py:58-114): BASELINE config 0 (Logo, Heat(50), order 30), Sensor(123)
      fixture of pygsp/tests/test_filters.py:12-29, the doctest value 0.27649 (filter.py:255-256)
  a7-a9  Filter.filter shape contract, synthesis, adjointness (filter.py:267-328)

Tolerances (BASELINE.json north_star): 1e-5 relative in float64, 1e-3 in float32, measured as
max|y - ref| / max|ref|; the tighter bounds asserted here flag indexing / scaling bugs early."""
import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_laplacian_4x4(golden_lap4, dtype):
    g = golden_lap4
    for name in ("und", "dir", "iso", "loop"):
        for lt in ("combinatorial", "normalized"):
            G = graphs.Graph(g["W_" + name], lap_type=lt, compute_dtype=dtype)
            ref = g["L_{}_{}".format(name, lt)]
            L = G.L
            assert L.has_canonical_format
            assert L.nnz == np.count_nonzero(ref), (name, lt)
            np.testing.assert_allclose(L.toarray(), ref, rtol=0,
                                       atol=1e-15 if dtype == np.float64 else 1e-6)
            np.testing.assert_allclose(G.dw, g["dw_" + name], rtol=1e-6)
            assert G.is_directed() == (name == "dir")
    with pytest.raises(ValueError):
        graphs.Graph(g["W_und"], lap_type="fancy")


def test_laplacian_sensor_logo_bit_level(golden_sensor123, golden_logo):
    W = csr_from(golden_sensor123, "W")
    for lt, key in (("combinatorial", "Lcomb"), ("normalized", "Lnorm")):
        dev = engine.DeviceGraph.from_w(W, lt)
        L, ref = dev.download_l(), csr_from(golden_sensor123, key)
        assert L.nnz == ref.nnz and dev.nnz_l == ref.nnz
        np.testing.assert_array_equal(L.indptr, ref.indptr)
        np.testing.assert_array_equal(L.indices, ref.indices)
        assert abs(L - ref).max() < 1e-15
        assert dev.nnz_internal % 4 == 0 and dev.nnz_internal >= ref.nnz
    # sequential row sums == scipy's column sums for an exactly symmetric W: bit-identical dw
    np.testing.assert_array_equal(engine.DeviceGraph.from_w(W).download_dw(), golden_sensor123["dw"])
    Wl = csr_from(golden_logo, "W")
    devl = engine.DeviceGraph.from_w(Wl)
    assert abs(devl.download_l() - csr_from(golden_logo, "L")).max() == 0
    np.testing.assert_array_equal(devl.download_dw(), golden_logo["dw"])


def test_empty_and_selfloop_graphs():
    """pygsp/tests/test_graphs.py:432-461: L.nnz == 0 for empty / self-loop-only graphs."""
    for W in (np.zeros((6, 6)), np.identity(6)):
        for lt in ("combinatorial", "normalized"):
            G = graphs.Graph(W, lap_type=lt)
            assert G.L.nnz == 0
            assert G.L.shape == (6, 6)


def test_adjacency_types():
    """pygsp/tests/test_graphs.py:464-485: float32 / int / CSC / COO / dense inputs."""
    W = random_graph(50, 6, 3)
    ref = orc.laplacian(W).toarray()
    for conv in (sparse.csc_matrix, sparse.coo_matrix, sparse.lil_matrix, lambda m: m.toarray()):
        G = graphs.Graph(conv(W))
        np.testing.assert_allclose(G.L.toarray(), ref, atol=1e-14)
    Wi = sparse.csr_matrix((W > 0).astype(np.int64))
    np.testing.assert_allclose(graphs.Graph(Wi).L.toarray(), orc.laplacian(Wi.astype(float)).toarray())
    G32 = graphs.Graph(W.astype(np.float32), compute_dtype=np.float32)
    np.testing.assert_allclose(G32.L.toarray(), ref, atol=1e-5)
    with pytest.raises(ValueError):
        graphs.Graph(np.ones((3, 4)))
    bad = W.toarray()
    bad[0, 1] = np.nan
    with pytest.raises(ValueError):
        graphs.Graph(bad)
    bad[0, 1] = np.inf
    with pytest.raises(ValueError):
        graphs.Graph(bad)


def test_capi_rejects_non_canonical_csr(ctx):
    indptr = np.array([0, 2, 2], dtype=np.int32)
    indices = np.array([1, 0], dtype=np.int32)  # not ascending
    data = np.ones(2)
    import ctypes
    h = ctypes.c_void_p()
    rc = _capi.load().gspx_graph_create_from_w(ctx._h, 2, 2, _capi.ptr(indptr), _capi.ptr(indices),
                                               _capi.ptr(data), _capi.F64, 0, _capi.F64, None,
                                               ctypes.byref(h))
    assert rc == _capi.ERR_INVALID and "canonical" in _capi.last_error()
    perm = np.array([0, 0], dtype=np.int32)  # not a permutation
    indices2 = np.array([0, 1], dtype=np.int32)
    rc = _capi.load().gspx_graph_create_from_w(ctx._h, 2, 2, _capi.ptr(indptr), _capi.ptr(indices2),
                                               _capi.ptr(data), _capi.F64, 0, _capi.F64,
                                               _capi.ptr(perm), ctypes.byref(h))
    assert rc == _capi.ERR_INVALID and "permutation" in _capi.last_error()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_logo_heat50_config0(golden_logo, dtype):
    """BASELINE.json configs[0]: Logo + Heat(50), 3 deltas, order 30."""
    g = golden_logo
    G = graphs.Graph(csr_from(g, "W"), compute_dtype=dtype)
    for tag in ("bounds", "fourier"):
        G._lmax = float(g["lmax_" + tag])
        h = filters.Heat(G, scale=50)
        np.testing.assert_allclose(filters.compute_cheby_coeff(h, m=30), g["coeff_" + tag],
                                   rtol=1e-12, atol=1e-16)
        y = h.filter(g["signal"], method="chebyshev", order=30)
        assert y.shape == (G.N,) and y.dtype == np.float64
        assert rel_err(y, g["y_" + tag]) < TOL[np.dtype(dtype)]
    G.estimate_lmax("bounds")
    assert abs(G.lmax / float(g["lmax_bounds"]) - 1) < (1e-12 if dtype == np.float64 else 1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sensor123_golden(golden_sensor123, dtype):
    g = golden_sensor123
    tol = TOL[np.dtype(dtype)]
    G = graphs.Graph(csr_from(g, "W"), coords=g["coords"], compute_dtype=dtype)
    G._lmax = float(g["lmax"])
    h = filters.Heat(G, scale=10)
    assert rel_err(h.filter(g["signal"], order=30), g["heat10_y"]) < tol
    assert rel_err(h.filter(g["signals5"], order=30), g["heat10_y5"]) < tol
    assert rel_err(h.filter(g["signal"], order=1), g["heat10_order1"]) < tol
    assert rel_err(h.filter(g["signal"], order=2), g["heat10_order2"]) < tol
    if dtype == np.float64:
        # test_approximations (test_filters.py:403-414): Chebyshev-30 == exact, rtol 1e-7
        np.testing.assert_allclose(h.filter(g["signal"], order=30), g["heat10_exact"], rtol=1e-7)
    with pytest.raises(TypeError):
        h.filter(g["signal"], order=0)
    with pytest.raises(ValueError):
        h.filter(g["signal"], method="lanczos")
    mh = filters.MexicanHat(G, Nf=6)
    np.testing.assert_allclose(np.array(filters.compute_cheby_coeff(mh, m=40)), g["mh6_c"],
                               rtol=1e-12, atol=1e-15)
    a = mh.filter(g["signals5"], order=40)
    assert a.shape == (123, 5, 6)
    assert rel_err(a, g["mh6_analysis"]) < tol
    assert rel_err(mh.filter(g["mh6_analysis"], order=40), g["mh6_synthesis"]) < tol
    a1 = mh.filter(g["signal"], order=40)
    assert a1.shape == (123, 6)
    assert rel_err(a1, g["mh6_analysis1"]) < tol
    assert rel_err(mh.filter(g["mh6_analysis1"], order=40), g["mh6_synthesis1"]) < tol
    # frame: the identity as signal, Nsig = N = 123 (test_filters.py:157-168, 82-84)
    h2 = filters.Heat(G, scale=[8, 9])
    F = h2.compute_frame(method="chebyshev", order=30)
    assert F.shape == (246, 123)
    assert rel_err(F, g["heat89_frame"]) < tol
    np.testing.assert_allclose(F.dot(g["signal"]).reshape(2, 123).T, h2.filter(g["signal"], order=30),
                               rtol=1e-7 if dtype == np.float64 else 1e-3, atol=1e-12)
    # normalized Laplacian with the bound lmax = 2
    G.compute_laplacian("normalized")
    G.estimate_lmax("bounds")
    assert G.lmax == 2
    hn = filters.Heat(G, scale=10)
    assert rel_err(hn.filter(g["signal"], order=30), g["heat10_norm_y"]) < tol


def test_doctest_value(golden_doctest):
    """filter.py:232-256 pinned number: ||s1 - s2|| = 0.27649."""
    g = golden_doctest
    G = graphs.Graph(csr_from(g, "W"))
    G._lmax = float(g["lmax"])
    s1 = np.zeros(G.N)
    s1[13] = 1
    s1 = filters.Heat(G, 3).filter(s1)
    assert rel_err(s1, g["s1"]) < 1e-12
    mh = filters.MexicanHat(G, Nf=4)
    s2 = mh.analyze(s1)
    assert s2.shape == (G.N, 4)
    s3 = mh.synthesize(s2)
    assert "{:.5f}".format(np.linalg.norm(s1 - s3)) == "0.27649"


def test_shape_contract():
    """Truth table of Filter.filter (pygsp/tests/test_filters.py:87-122)."""
    N = 40
    W = random_graph(N, 5, 11)
    G = graphs.Graph(W)
    G.estimate_lmax("bounds")
    rng = np.random.default_rng(0)
    heat, mh = filters.Heat(G), filters.MexicanHat(G, Nf=6)
    for shape, out1, out6 in [((N,), (N,), (N, 6)), ((N, 1), (N,), (N, 6)), ((N, 1, 1), (N,), (N, 6)),
                              ((N, 4), (N, 4), (N, 4, 6)), ((N, 4, 1), (N, 4), (N, 4, 6)),
                              ((N, 6), (N, 6), (N,)), ((N, 4, 6), None, (N, 4)),
                              ((N, 6, 6), None, (N, 6)), ((N, 6, 1), (N, 6), (N, 6, 6)),
                              ((N, 4, 3), None, None), ((N, 0), (N, 0), (N, 0, 6))]:
        s = rng.standard_normal(shape)
        for f, expect in ((heat, out1), (mh, out6)):
            if expect is None:
                with pytest.raises(ValueError):
                    f.filter(s)
            else:
                assert f.filter(s).shape == expect, (shape, f.Nf)
    with pytest.raises(ValueError):
        heat.filter(rng.standard_normal((N, 2, 2, 1)))
    with pytest.raises(ValueError):
        heat.filter(rng.standard_normal(N + 1))
    with pytest.raises(ValueError):
        mh.analyze(rng.standard_normal((N, 4, 6)))
    with pytest.raises(ValueError):
        mh.synthesize(rng.standard_normal((N, 4)))
    # integer / list / Fortran-ordered / float32 inputs give float64 C-ordered results
    base = rng.integers(-3, 4, size=(N, 3))
    ref = heat.filter(base.astype(float))
    for s in (base, base.tolist(), np.asfortranarray(base.astype(float)), base.astype(np.float32)):
        y = heat.filter(s)
        assert y.dtype == np.float64
        np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-14)
    # inputs are never mutated
    s = rng.standard_normal((N, 3))
    keep = s.copy()
    mh.filter(s)
    np.testing.assert_array_equal(s, keep)
    # localize = sqrt(N) * filter(delta_i)
    d = np.zeros(N)
    d[7] = 1
    np.testing.assert_allclose(heat.localize(7), np.sqrt(N) * heat.filter(d))


# ---------------------------------------------------------------------------------------------
# seeded random graphs vs the oracle: every panel width, both kernels, both dtypes
# ---------------------------------------------------------------------------------------------
NSIGS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 31, 32, 33, 64, 65, 100, 128, 130, 257]


def test_host_laplacian_upload_matches_device_build(ctx):
    """gspx_graph_create_from_l (bit-parity mode) == device-built Laplacian."""
    W = random_graph(2000, 8, seed=9)
    L = orc.laplacian(W, "normalized")
    x = np.random.default_rng(2).standard_normal((2000, 6))
    c = orc.compute_cheby_coeff(orc.heat_kernel(3, 2.0), 2.0, 20)
    a = engine.DeviceGraph.from_w(W, "normalized", ctx=ctx)
    b = engine.DeviceGraph.from_l(L, ctx=ctx)
    ya, _ = a.cheby_filter(c, x, 2.0)
    yb, _ = b.cheby_filter(c, x, 2.0)
    ref = orc.cheby_op(L, 2.0, c, x)
    assert rel_err(ya[0], ref) < 1e-12 and rel_err(yb[0], ref) < 1e-12
    with pytest.raises(ValueError):
        b.download_dw()


def test_analysis_synthesis_adjoint():
    """<g.filter(x), y> == <x, g.filter(y)>."""
    G = graphs.Sensor(500, seed=3)
    G.estimate_lmax("bounds")
    mh = filters.MexicanHat(G, Nf=5)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((G.N, 3))
    y = rng.standard_normal((G.N, 3, 5))
    lhs = np.sum(mh.filter(x, order=25) * y)
    rhs = np.sum(x * mh.filter(y, order=25))
    assert abs(lhs - rhs) < 1e-11 * max(abs(lhs), 1.0)
