"""The opt-in Newton-form evaluation of the same polynomial (gspx_newton_filter*; no reference
counterpart - the reference's result is the bar) and its two-orders-per-pass kernel.  `-m gpu`."""
import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nsig", [1, 3, 8, 16, 33, 64, 130])
def test_newton_form_matches_reference(ctx, golden_sensor123, dtype, nsig):
    n = 3001
    W = random_graph(n, 9, seed=200 + nsig, hub=True, isolated=4)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    rng = np.random.default_rng(nsig)
    x = rng.standard_normal((n, nsig))
    perm = rng.permutation(n).astype(np.int32) if nsig % 2 else None
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
    tol = TOL[np.dtype(dtype)] * 10
    for scale, order in ((7, 25), (40, 30), (5, 1), (5, 2)):
        c = orc.compute_cheby_coeff(orc.heat_kernel(scale, lmax), lmax, order)
        nodes, d = filters.cheb_to_newton(c)
        y, ms = dev.newton_filter(nodes, d, x, lmax)
        ref = orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64))
        assert rel_err(y, ref) < tol, (scale, order)
    for kern in (1, 5):
        if nsig <= 4:
            break
        ctx.set_option("kernel", kern)
        y, _ = dev.newton_filter(nodes, d, x, lmax)
        assert rel_err(y, ref) < tol, kern
    ctx.set_option("kernel", 0)
    ctx.set_option("max_batch", 8)
    y, _ = dev.newton_filter(nodes, d, x, lmax)
    assert rel_err(y, ref) < tol
    ctx.set_option("max_batch", 0)
    dev.destroy()


def test_newton_form_through_filter_api(golden_sensor123, golden_logo):
    g = golden_sensor123
    G = graphs.Graph(csr_from(g, "W"), coords=g["coords"])
    G._lmax = float(g["lmax"])
    h = filters.Heat(G, scale=10)
    try:
        filters.set_evaluation("newton")
        assert rel_err(h.filter(g["signal"], order=30), g["heat10_y"]) < 1e-12
        assert rel_err(h.filter(g["signals5"], order=30), g["heat10_y5"]) < 1e-12
        # filterbanks and synthesis keep the recurrence
        mh = filters.MexicanHat(G, Nf=6)
        assert rel_err(mh.filter(g["signals5"], order=40), g["mh6_analysis"]) < 1e-11
        assert rel_err(mh.filter(g["mh6_analysis"], order=40), g["mh6_synthesis"]) < 1e-11
        Gl = graphs.Graph(csr_from(golden_logo, "W"))
        Gl._lmax = float(golden_logo["lmax_bounds"])
        y = filters.Heat(Gl, 50).filter(golden_logo["signal"], order=30)
        assert rel_err(y, golden_logo["y_bounds"]) < 1e-12
        with pytest.raises(TypeError):
            h.filter(g["signal"], order=0)
    finally:
        filters.set_evaluation("recurrence")
    assert rel_err(filters.cheby_op(G, filters.compute_cheby_coeff(h, m=30), g["signal"],
                                    evaluation="newton"), g["heat10_y"]) < 1e-12
