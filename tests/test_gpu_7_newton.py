"""The Newton-form evaluation of the same polynomial (gspx_newton_filter*; evaluation='newton' / 'auto'; no reference
counterpart - the reference's result is the bar): small graphs over widths and dtypes, the drop-in API, and the
headline size over orders 30-200 and the Heat / Mexican-hat kernels behind the host-side guard.  `-m gpu`."""
import numpy as np
import pytest

from conftest import csr_from, rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import engine, filters, graphs

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


HEADLINE = {}


def headline_case():
    """Sensor(1M, k = 8) and 64 signals whose values are exact in both compute dtypes (one oracle serves both), the
    oracle's T_k-sharing call per order for the eight kernels: two Heat scales and the six Mexican-hat bands."""
    if HEADLINE:
        return HEADLINE
    G = graphs.Sensor(1000000, k=8, seed=42)
    G.estimate_lmax("bounds")
    x = np.random.default_rng(7).standard_normal((G.N, 64)).astype(np.float32)
    banks = [filters.Heat(G, 50), filters.Heat(G, 10)] + [filters.MexicanHat(G, Nf=6)[i] for i in range(6)]
    names = ["heat50", "heat10"] + ["mexican_hat_band%d" % i for i in range(6)]
    L = G.L.astype(np.float64)
    ref = {}
    for order in (30, 50, 100, 200):
        C = np.array([filters.compute_cheby_coeff(b, m=order) for b in banks])
        ref[order] = orc.cheby_op(L, G.lmax, C, x[:, :2].astype(np.float64)).reshape(len(banks), G.N, 2)
    HEADLINE.update(G=G, x=x, banks=banks, names=names, ref=ref)
    return HEADLINE


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_auto_evaluation_at_headline_size(dtype):
    """VERDICT r5 "Next 2b": evaluation='auto' on the headline graph, 1M vertices x 64 signals, orders 30 / 50 / 100 /
    200, Heat(50), Heat(10) and every single Mexican-hat band, device resident, against the oracle at a TENTH of the
    bar.  'auto' runs the Newton form wherever the guard clears the polynomial for the dtype (every case in fp64; fp32
    up to order 100) and the recurrence elsewhere (fp32 at order 200: the Newton coefficients overflow) - an explicit
    'newton' raises there instead of computing garbage."""
    h = headline_case()
    G, x = h["G"], h["x"]
    dt = np.dtype(dtype)
    tol = BAR[dt] / 10
    dev = G.device_graph(dtype)
    xd = engine.DeviceArray.from_host(dev.ctx, x, dtype)
    old = G.compute_dtype
    G.compute_dtype = dt
    worst, picked = {}, {}
    try:
        for order in (30, 50, 100, 200):
            for i, (name, bank) in enumerate(zip(h["names"], h["banks"])):
                c = filters.compute_cheby_coeff(bank, m=order)
                expect = ("product" if filters.product_guard(c, dt)[0] else
                          "newton" if filters.newton_guard(filters.effective_coefficients(c, dt), dt)[0] else "recurrence")
                assert expect == filters.choose_evaluation("auto", np.atleast_2d(c), dt, G.N, 64)
                picked[expect] = picked.get(expect, 0) + 1
                y = bank.filter(xd, order=order, evaluation="auto")
                assert isinstance(y, engine.DeviceArray) and G._gspx_last_evaluation == expect
                got = np.asarray(y)[:, :2].astype(np.float64)
                err = rel_err(got, h["ref"][order][i])
                worst[(name, order)] = err
                assert err < tol, (name, order, expect, err)
                del y
        if dt == np.float32:
            with pytest.raises(ValueError, match="not representable"):
                h["banks"][0].filter(xd, order=200, evaluation="newton")
        # 'auto' == the form it names, bit for bit; every form on its own at a tenth of the bar; the recurrence stays
        # the default
        assert picked.get("product", 0) >= 16 and picked.get("newton", 0) >= 2, picked
        c30 = filters.compute_cheby_coeff(h["banks"][0], m=30)
        how = "product" if filters.product_guard(c30, dt)[0] else "newton"
        y_auto = np.asarray(h["banks"][0].filter(xd, order=30, evaluation="auto"))
        assert np.array_equal(y_auto, np.asarray(h["banks"][0].filter(xd, order=30, evaluation=how)))
        for form in ("newton", "product"):
            yf = np.asarray(h["banks"][0].filter(xd, order=30, evaluation=form))[:, :2].astype(np.float64)
            assert rel_err(yf, h["ref"][30][0]) < tol, form
        h["banks"][0].filter(xd, order=30)
        assert G._gspx_last_evaluation == "recurrence"
    finally:
        G.compute_dtype = old
        xd.free()
        if dt != old:
            G._dev.pop(dt).destroy()
    print("worst auto error at 1M x 64, {}: {:.2e} (bar/10 = {:.0e}); forms picked: {}".format(
        dt.name, max(worst.values()), tol, picked))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_auto_evaluation_on_degenerate_polynomials(dtype):
    """evaluation='auto' where the polynomial is not a filter's: the zero polynomial, a constant, degree one, the
    shortest coefficient vector the reference accepts, an alternating series (no decay: nothing to trim), a short heat
    series, random coefficients of order 200 - whatever form is picked, the result is the recurrence's to a tenth of
    the bar and finite."""
    h = headline_case()
    G = h["G"]
    dt = np.dtype(dtype)
    x = h["x"][:, :16]
    rng = np.random.default_rng(11)
    cases = {
        "zero": np.zeros(31), "constant": np.r_[2.0, np.zeros(30)], "degree one": np.r_[2.0, 1.0, np.zeros(29)],
        "two coefficients": np.array([1.0, 0.5]), "alternating": (-1.0) ** np.arange(31),
        "heat order 5": orc.compute_cheby_coeff(orc.heat_kernel(10, G.lmax), G.lmax, 5),
        "random order 200": rng.standard_normal(201),
    }
    old = G.compute_dtype
    G.compute_dtype = dt
    try:
        for name, c in cases.items():
            y_rec = filters.cheby_op(G, c, x.astype(dtype), evaluation="recurrence")
            y_auto = filters.cheby_op(G, c, x.astype(dtype), evaluation="auto")
            how = G._gspx_last_evaluation
            assert how == filters.choose_evaluation("auto", np.atleast_2d(c), dt, G.N, 16), name
            assert np.all(np.isfinite(y_auto)), (name, how)
            den = max(float(np.max(np.abs(y_rec))), 1e-300)
            assert float(np.max(np.abs(y_auto - y_rec))) / den < BAR[dt] / 10, (name, how)
            if name in ("zero", "constant"):
                assert np.array_equal(y_auto, y_rec), (name, how)  # 0 and c_0 / 2 * x: exact in every form
    finally:
        G.compute_dtype = old
        if dt != old:
            G._dev.pop(dt).destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nsig", [1, 3, 8, 16, 33, 64, 130])
def test_product_form_matches_reference(ctx, dtype, nsig):
    """gspx_poly_program in product form (filters.cheb_to_product) against the oracle on ragged graphs with a hub row
    and isolated vertices, every panel width, with and without an internal order, the plain kernels and batching."""
    n = 3001
    W = random_graph(n, 9, seed=300 + nsig, hub=True, isolated=4)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    rng = np.random.default_rng(nsig)
    x = rng.standard_normal((n, nsig))
    perm = rng.permutation(n).astype(np.int32) if nsig % 2 else None
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
    tol = TOL[np.dtype(dtype)] * 100
    for scale, order in ((7, 25), (40, 30), (5, 1), (5, 2), (3, 60)):
        c = orc.compute_cheby_coeff(orc.heat_kernel(scale, lmax), lmax, order)
        if not filters.product_guard(c, dtype)[0]:
            continue
        prog = filters.cheb_to_product(c, dtype)
        y, ms = dev.program_filter(prog, x, lmax)
        ref = orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64))
        assert rel_err(y, ref) < tol, (scale, order)
    for kern in (1, 5):
        if nsig <= 4:
            break
        ctx.set_option("kernel", kern)
        y, _ = dev.program_filter(prog, x, lmax)
        assert rel_err(y, ref) < tol, kern
    ctx.set_option("kernel", 0)
    ctx.set_option("max_batch", 8)
    y, _ = dev.program_filter(prog, x, lmax)
    assert rel_err(y, ref) < tol
    ctx.set_option("max_batch", 0)
    # the Newton form is the same entry point with old_is_x: equal to gspx_newton_filter bit for bit
    c = orc.compute_cheby_coeff(orc.heat_kernel(7, lmax), lmax, 25)
    nodes, d = filters.cheb_to_newton(c)
    K = nodes.size
    rows = [[0.5 * d[K], 0.0, d[K - 1] - d[K] * nodes[K - 1]]] + [[0.5, -nodes[K - 1 - s], d[K - 1 - s]] for s in range(1, K)]
    y_prog, _ = dev.program_filter(np.array(rows), x, lmax, old_is_x=True)
    y_newt, _ = dev.newton_filter(nodes, d, x, lmax)
    assert np.array_equal(y_prog, y_newt)
    with pytest.raises(TypeError):
        dev.program_filter(np.zeros((0, 3)), x, lmax)
    dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_programs_on_host_arrays_are_pipelined_like_the_recurrence(ctx, dtype):
    """gspx_poly_program with host arrays large enough to pipeline (column batches over pinned staging, the program on
    every batch): bit-identical to the one-shot form (host_pipeline = 0), for the product and the Newton program, and
    right against the oracle; the stage times of the pipelined call are reported like the recurrence's."""
    G = graphs.Sensor(200000, k=8, seed=11, compute_dtype=dtype, ctx=ctx)
    G.estimate_lmax("bounds")
    dev = G.device_graph(dtype)
    x = np.random.default_rng(5).standard_normal((G.N, 64)).astype(dtype)  # 51 / 102 MB: pipelined
    c = orc.compute_cheby_coeff(orc.heat_kernel(10, G.lmax), G.lmax, 40)
    ref = orc.cheby_op(orc.laplacian(G.W), G.lmax, c, x[:, :2].astype(np.float64))
    progs = {"product": (filters.cheb_to_product(c, dtype), False),
             "newton": (filters.newton_program(*filters.cheb_to_newton(c)), True)}
    for name, (prog, old_is_x) in progs.items():
        ctx.set_option("host_pipeline", 2)
        y_pipe, _ = dev.program_filter(prog, x, G.lmax, old_is_x=old_is_x)
        stages = ctx.last_host_timing()
        assert stages is not None and stages["batches"] >= 2, name
        ctx.set_option("host_pipeline", 0)
        y_one, _ = dev.program_filter(prog, x, G.lmax, old_is_x=old_is_x)
        assert ctx.last_host_timing() is None
        ctx.set_option("host_pipeline", 1)
        assert np.array_equal(y_pipe, y_one), name
        assert rel_err(y_pipe[:, :2], ref) < BAR[np.dtype(dtype)] / 10, name
    # through the API: evaluation='auto' on numpy arrays
    y = filters.Heat(G, 10).filter(x, order=40, evaluation="auto")
    assert G._gspx_last_evaluation == "product" and rel_err(y[:, :2], ref) < BAR[np.dtype(dtype)] / 10
