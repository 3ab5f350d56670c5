"""The oracle (oracle/cheby_oracle.py) pinned against golden vectors produced by the real
reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import csr_from, rel_err
from oracle import cheby_oracle as orc


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
