"""The oracle (oracle/cheby_oracle.py) pinned against golden vectors produced by the real
reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, csr_from, rel_err
from oracle import cheby_oracle as orc
from oracle import knn_oracle as knn
from oracle import ops_oracle as ops


def test_laplacians_4x4(golden_lap4):
    g = golden_lap4
    for name in ("und", "dir", "iso", "loop"):
        W = g["W_" + name]
        for lt in ("combinatorial", "normalized"):
            L = orc.laplacian(W, lt).toarray()
            np.testing.assert_allclose(L, g["L_{}_{}".format(name, lt)], rtol=0, atol=1e-15)
        np.testing.assert_allclose(orc.degree(W), g["dw_" + name], rtol=0, atol=0)


def test_laplacian_unknown_type():
    with pytest.raises(ValueError):
        orc.laplacian(np.zeros((3, 3)), "fancy")


def test_laplacian_sensor_and_logo(golden_sensor123, golden_logo):
    W = csr_from(golden_sensor123, "W")
    for lt, key in (("combinatorial", "Lcomb"), ("normalized", "Lnorm")):
        L = orc.laplacian(W, lt)
        Lref = csr_from(golden_sensor123, key)
        assert L.nnz == Lref.nnz
        assert abs(L - Lref).max() < 1e-15
    Wl = csr_from(golden_logo, "W")
    assert abs(orc.laplacian(Wl) - csr_from(golden_logo, "L")).max() == 0
    np.testing.assert_array_equal(orc.degree(Wl), golden_logo["dw"])


def test_coefficients(golden_logo, golden_sensor123):
    for tag in ("bounds", "fourier"):
        lmax = float(golden_logo["lmax_" + tag])
        c = orc.compute_cheby_coeff(orc.heat_kernel(50, lmax), lmax, 30)
        np.testing.assert_allclose(c, golden_logo["coeff_" + tag], rtol=1e-13, atol=1e-16)
    # values quoted in SURVEY.md 8c
    assert abs(float(golden_logo["lmax_bounds"]) - 18.583333333333332) < 1e-12
    np.testing.assert_allclose(golden_logo["coeff_bounds"][:3],
                               [0.160393547095, -0.157152226639, 0.147821368964], rtol=1e-9)
    lmax = float(golden_sensor123["lmax"])
    cm = np.array([orc.compute_cheby_coeff(k, lmax, 40) for k in orc.mexican_hat_kernels(lmax, 6)])
    np.testing.assert_allclose(cm, golden_sensor123["mh6_c"], rtol=1e-12, atol=1e-15)


def test_logo_heat50(golden_logo):
    """BASELINE.json configs[0]."""
    L = csr_from(golden_logo, "L")
    s = golden_logo["signal"]
    for tag in ("bounds", "fourier"):
        lmax = float(golden_logo["lmax_" + tag])
        y = orc.filter_chebyshev(L, lmax, [orc.heat_kernel(50, lmax)], s, 30)
        assert rel_err(y, golden_logo["y_" + tag]) < 1e-14
    yb = golden_logo["y_bounds"]
    assert abs(yb.sum() - 2.9999999797598558) < 1e-12
    assert abs(np.linalg.norm(yb) - 0.16375128493237434) < 1e-13


def test_sensor123_all(golden_sensor123):
    g = golden_sensor123
    L = csr_from(g, "Lcomb")
    lmax = float(g["lmax"])
    heat = [orc.heat_kernel(10, lmax)]
    assert rel_err(orc.filter_chebyshev(L, lmax, heat, g["signal"], 30), g["heat10_y"]) < 1e-14
    assert rel_err(orc.filter_chebyshev(L, lmax, heat, g["signals5"], 30), g["heat10_y5"]) < 1e-14
    assert rel_err(orc.filter_chebyshev(L, lmax, heat, g["signal"], 1), g["heat10_order1"]) < 1e-14
    assert rel_err(orc.filter_chebyshev(L, lmax, heat, g["signal"], 2), g["heat10_order2"]) < 1e-14
    # test_approximations (test_filters.py:403-414): Chebyshev-30 == exact at rtol 1e-7
    np.testing.assert_allclose(orc.filter_chebyshev(L, lmax, heat, g["signal"], 30), g["heat10_exact"],
                               rtol=1e-7)
    mh = orc.mexican_hat_kernels(lmax, 6)
    a = orc.filter_chebyshev(L, lmax, mh, g["signals5"], 40)
    assert a.shape == (123, 5, 6)
    assert rel_err(a, g["mh6_analysis"]) < 1e-13
    assert rel_err(orc.filter_chebyshev(L, lmax, mh, a, 40), g["mh6_synthesis"]) < 1e-13
    a1 = orc.filter_chebyshev(L, lmax, mh, g["signal"], 40)
    assert rel_err(a1, g["mh6_analysis1"]) < 1e-13
    assert rel_err(orc.filter_chebyshev(L, lmax, mh, a1, 40), g["mh6_synthesis1"]) < 1e-13
    # frame (test_filters.py:157-168 uses compute_frame(chebyshev, order=30))
    two = [orc.heat_kernel(8, lmax), orc.heat_kernel(9, lmax)]
    F = orc.filter_chebyshev(L, lmax, two, np.identity(123), 30).T.reshape(-1, 123)
    assert rel_err(F, g["heat89_frame"]) < 1e-13
    # normalized Laplacian, lmax = 2 bound
    Ln = csr_from(g, "Lnorm")
    ln = float(g["lmax_norm"])
    assert ln == 2.0
    assert rel_err(orc.filter_chebyshev(Ln, ln, [orc.heat_kernel(10, ln)], g["signal"], 30),
                   g["heat10_norm_y"]) < 1e-14


def test_order0_raises(golden_sensor123):
    L = csr_from(golden_sensor123, "Lcomb")
    with pytest.raises(TypeError):
        orc.cheby_op(L, 10.0, np.array([1.0]), np.ones(123))


def test_doctest_value(golden_doctest):
    """filter.py:255-256: ||s1 - s2|| = 0.27649."""
    g = golden_doctest
    assert "{:.5f}".format(float(g["norm"])) == "0.27649"
    L = orc.laplacian(csr_from(g, "W"))
    lmax = float(g["lmax"])
    mh = orc.mexican_hat_kernels(lmax, 4)
    s2 = orc.filter_chebyshev(L, lmax, mh, g["s1"], 30)
    s3 = orc.filter_chebyshev(L, lmax, mh, s2, 30)
    assert rel_err(s2, g["s2"]) < 1e-13
    assert "{:.5f}".format(np.linalg.norm(g["s1"] - s3)) == "0.27649"


def test_cheby_rect(golden_sensor123):
    g = golden_sensor123
    L, lmax = csr_from(g, "Lcomb"), float(g["lmax"])
    assert rel_err(orc.cheby_rect(L, lmax, g["rect_bounds"], g["signal"], 30), g["rect_y"]) < 1e-14
    assert rel_err(orc.cheby_rect(L, lmax, g["rect_bounds"], g["signals5"], 25), g["rect_y5"]) < 1e-14


# ---- the operators next to the path (oracle/ops_oracle.py) -------------------------
def test_ops_doctest_values():
    """graph.py:686-699: Path(5), signal [0, 2, 2, 4, 4]."""
    from scipy import sparse
    W = sparse.diags([np.ones(4), np.ones(4)], [1, -1]).tocsr()
    x = np.array([0, 2, 2, 4, 4.0])
    assert ops.dirichlet_energy(ops.laplacian(W), x) == 8.0
    D = ops.differential_operator(W)
    np.testing.assert_allclose(ops.grad(D, x), [2, 0, 2, 0])
    Wd = sparse.diags([np.ones(4)], [1]).tocsr()  # directed path
    assert abs(ops.dirichlet_energy(ops.laplacian(Wd), x) - 4.0) < 1e-15
    np.testing.assert_allclose(ops.grad(ops.differential_operator(Wd), x),
                               [1.41421356, 0, 1.41421356, 0], atol=1e-8)
    # difference.py:104-118: the 3-vertex examples
    W3 = np.array([[0, 2, 0], [2, 0, 1], [0, 1, 0.0]])
    np.testing.assert_allclose(ops.differential_operator(W3).toarray(),
                               [[-1.41421356, 0], [1.41421356, -1], [0, 1]], atol=1e-8)
    np.testing.assert_allclose(ops.differential_operator(W3, "normalized").toarray(),
                               [[-1, 0], [0.81649658, -0.57735027], [0, 1]], atol=1e-8)
    W3d = np.array([[0, 2, 0], [2, 0, 1], [0, 0, 0.0]])
    np.testing.assert_allclose(ops.differential_operator(W3d).toarray(),
                               [[-1, 1, 0], [1, -1, -0.70710678], [0, 0, 0.70710678]], atol=1e-8)


def test_ops_sensor123(golden_ops):
    g = golden_ops
    W = csr_from(g, "W")
    x, X5 = g["x"], g["X5"]
    src, dst, w = ops.get_edge_list(W)
    np.testing.assert_array_equal(src, g["edges_src"])
    np.testing.assert_array_equal(dst, g["edges_dst"])
    np.testing.assert_array_equal(w, g["edges_w"])
    for lt in ("combinatorial", "normalized"):
        L = ops.laplacian(W, lt)
        D = ops.differential_operator(W, lt)
        np.testing.assert_allclose(D.toarray(), g["D_" + lt], rtol=0, atol=1e-15)
        assert abs(ops.dirichlet_energy(L, x) - float(g["energy_" + lt])) <= 1e-13 * abs(float(g["energy_" + lt]))
        assert rel_err(ops.dirichlet_energy(L, X5), g["energy5_" + lt]) < 1e-14
        assert rel_err(ops.grad(D, x), g["grad_" + lt]) < 1e-15
        assert rel_err(ops.grad(D, X5), g["grad5_" + lt]) < 1e-15
        assert rel_err(ops.div(D, ops.grad(D, x)), g["div_" + lt]) < 1e-15
        assert rel_err(ops.div(D, ops.grad(D, X5)), g["div5_" + lt]) < 1e-15
        assert rel_err(L.dot(X5), g["Lx_" + lt]) < 1e-15
        # L = D D^T (difference.py:31)
        assert abs(D.dot(D.T) - L).max() < 1e-13


def test_ops_directed_and_self_loops():
    """difference.py:160-166 through the oracle against the real reference (tests/golden/ops_directed.npz): a
    directed weighted graph (every stored entry an edge, D / sqrt(2)) and an undirected one with self-loops."""
    g = load_golden("ops_directed.npz")
    x, X4 = g["x"], g["X4"]
    for name in ("dir", "loops"):
        W = csr_from(g, "W" + name)
        src, dst, w = ops.get_edge_list(W)
        np.testing.assert_array_equal(src, g["src_" + name])
        np.testing.assert_array_equal(dst, g["dst_" + name])
        np.testing.assert_array_equal(w, g["w_" + name])
        assert src.size == int(g["ne_" + name]) and np.all(np.diff(src) >= 0)
        np.testing.assert_allclose(ops.degree(W), g["dw_" + name], rtol=1e-15)
        for lt in ("combinatorial", "normalized"):
            key = "{}_{}".format(name, lt)
            D = ops.differential_operator(W, lt)
            np.testing.assert_allclose(D.toarray(), g["D_" + key], rtol=0, atol=1e-15)
            np.testing.assert_allclose(ops.laplacian(W, lt).toarray(), g["L_" + key], rtol=0, atol=1e-15)
            assert rel_err(ops.grad(D, x), g["grad_" + key]) < 1e-15
            assert rel_err(ops.grad(D, X4), g["grad4_" + key]) < 1e-15
            assert rel_err(ops.div(D, ops.grad(D, x)), g["div_" + key]) < 1e-15
            assert rel_err(ops.div(D, ops.grad(D, X4)), g["div4_" + key]) < 1e-15


def test_ops_tikhonov(golden_ops):
    g = golden_ops
    L = ops.laplacian(csr_from(g, "W"))
    mask = g["mask"]
    m0 = np.nan_to_num(g["measures"])
    for tau in (0.5, 5.0):
        assert rel_err(ops.regression_tikhonov(L, m0.copy(), mask, tau), g["reg_tau%g" % tau]) < 1e-13
    assert rel_err(ops.regression_tikhonov(L, g["reg3_in"].copy(), mask, 0.5), g["reg3_tau0.5"]) < 1e-13
    assert rel_err(ops.regression_tikhonov(L, m0.copy(), mask, 0), g["reg_tau0"]) < 1e-12
    lab = g["labels"].astype(float)
    lab[~mask] = np.nan
    assert rel_err(ops.classification_tikhonov(L, lab.copy(), mask, 0.1), g["class_tau0.1"]) < 1e-13
    assert rel_err(ops.classification_tikhonov(L, lab.copy(), mask, 0), g["class_tau0"]) < 1e-12
    with pytest.raises(ValueError):
        ops.regression_tikhonov(L, m0, mask[:-1], 0)


# ---- nearest-neighbour graph construction (oracle/knn_oracle.py) ----------------------
def test_knn_golden(golden_knn):
    g = golden_knn
    X = knn.preprocess(g["X3"])
    np.testing.assert_allclose(X, g["X3_coords"], rtol=0, atol=1e-14)
    W, sigma, NN, D = knn.knn_weights(X, 5)
    Wref = csr_from(g, "W3")
    assert abs(sigma - float(g["sigma3"])) < 1e-15
    assert W.nnz == Wref.nnz and abs(W - Wref).max() < 1e-16
    assert NN.shape == (200, 5) and (np.diff(D, axis=1) >= 0).all()
    W1, s1, _, _ = knn.knn_weights(knn.preprocess(g["X1"], False, False), 3, sigma=0.7)
    assert s1 == 0.7 and abs(W1 - csr_from(g, "W1")).max() < 1e-16
    Ws, sg, _, _ = knn.knn_weights(knn.sensor_coords(123, seed=42), 6)
    np.testing.assert_array_equal(knn.sensor_coords(123, seed=42), g["sensor_coords"])
    assert abs(Ws - csr_from(g, "Wsensor")).max() < 1e-16 and abs(sg - float(g["sensor_sigma"])) < 1e-16
    cd = knn.sensor_coords(144, seed=7, distributed=True)
    np.testing.assert_array_equal(cd, g["dist_coords"])
    assert abs(knn.knn_weights(cd, 4)[0] - csr_from(g, "Wdist")).max() < 1e-16
    with pytest.raises(ValueError):
        knn.knn_weights(cd[:4], 4)


def test_knn_highdim_golden():
    """The oracle against the reference's NNGraph on 9-, 25-, 6- and 40-dimensional clouds (tests/golden/knn_highdim.npz)."""
    g = load_golden("knn_highdim.npz")
    for tag, k in (("p9", 8), ("p25", 10)):
        X = knn.preprocess(g["X_" + tag])
        np.testing.assert_allclose(X, g["coords_" + tag], rtol=0, atol=1e-13)
        W, sigma, NN, D = knn.knn_weights(X, k)
        Wref = csr_from(g, "W_" + tag)
        assert W.nnz == Wref.nnz and abs(W - Wref).max() < 1e-15 and abs(sigma - float(g["sigma_" + tag])) < 1e-14
    W, _, _, _ = knn.knn_weights(knn.preprocess(g["X6"]), 7, dist_type="manhattan")
    assert abs(W - csr_from(g, "W6_manhattan")).max() < 1e-15
    W, _, _, _ = knn.knn_weights(g["X6"], 5, symmetrize_type="maximum")
    assert abs(W - csr_from(g, "W6_maximum")).max() < 1e-15
    W, s, _, _ = knn.knn_weights(knn.preprocess(g["X40"]), 12, sigma=2.5)
    assert s == 2.5 and abs(W - csr_from(g, "W40")).max() < 1e-15
    W, sg = knn.radius_weights(knn.preprocess(g["X6"]), 0.125)
    Wref = csr_from(g, "W6_radius")
    assert W.nnz == Wref.nnz and abs(W - Wref).max() < 1e-15 and abs(sg - float(g["sigma6_radius"])) < 1e-15
    W, _ = knn.radius_weights(knn.preprocess(g["X_p9"]), 0.3, dist_type="manhattan")
    assert W.nnz == csr_from(g, "Wp9_radius_manhattan").nnz and abs(W - csr_from(g, "Wp9_radius_manhattan")).max() < 1e-15


def test_radius_golden(golden_knn):
    g = golden_knn
    W, sigma = knn.radius_weights(knn.preprocess(g["Xr"]), 0.35)
    Wref = csr_from(g, "Wr")
    assert W.nnz == Wref.nnz and abs(W - Wref).max() < 1e-16 and abs(sigma - float(g["sigma_r"])) < 1e-16
    W2, s2 = knn.radius_weights(g["X2r"], 0.08, sigma=0.01)
    assert s2 == 0.01 and W2.nnz == csr_from(g, "W2r").nnz and abs(W2 - csr_from(g, "W2r")).max() < 1e-16
    with pytest.raises(ValueError):
        knn.radius_weights(g["X2r"][:5] * 100, 0.01)  # nobody within reach: "No neighbors found"


def test_other_metrics_golden(golden_knn):
    g = golden_knn
    W = knn.knn_weights(knn.preprocess(g["X3"]), 6, dist_type="manhattan")[0]
    assert abs(W - csr_from(g, "W3_manhattan")).max() < 1e-16
    W = knn.knn_weights(g["X3"], 4, dist_type="max_dist")[0]
    assert abs(W - csr_from(g, "W3_maxdist")).max() < 1e-16
    W = knn.radius_weights(g["X2r"], 0.07, dist_type="manhattan")[0]
    assert W.nnz == csr_from(g, "W2r_manhattan").nnz and abs(W - csr_from(g, "W2r_manhattan")).max() < 1e-16


def test_symmetrize_types_golden(golden_knn):
    g = golden_knn
    X = knn.preprocess(g["X3"])
    for st in ("maximum", "fill", "tril", "triu"):
        W = knn.knn_weights(X, 5, symmetrize_type=st)[0]
        Wref = csr_from(g, "W3_" + st)
        Wref.eliminate_zeros()
        assert W.nnz == Wref.nnz and abs(W - Wref).max() < 1e-16, st
