"""Device-resident signals through the drop-in API (engine.DeviceArray) and the secondary seam of SURVEY 8(b):
``plugin.install(wrap_filter=True)`` replacing ``Filter.filter`` / ``Filter.compute_frame`` of a pygsp-shaped
package (filter.py:146-328, 506-600).  Real MI355X (`-m gpu`); the oracle is the checker."""
import types

import numpy as np
import pytest

from conftest import csr_from, rel_err
from gpu_helpers import random_graph, upper_lmax
from oracle import cheby_oracle as orc
from pygsp_amd import engine, filters, graphs

pytestmark = pytest.mark.gpu


def small_graph(n=300, seed=5, dtype=np.float64):
    G = graphs.Graph(random_graph(n, 6, seed), compute_dtype=dtype)
    G.estimate_lmax("bounds")
    return G


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_array_round_trip(dtype):
    """to_device / np.asarray: any real dtype and memory order in, the float64 array the reference's filter()
    would have been handed back out; shapes as given, including empty panels."""
    G = small_graph(dtype=dtype)
    rng = np.random.default_rng(0)
    tol = 0 if dtype == np.float64 else 1e-7
    for shape in [(G.N,), (G.N, 1), (G.N, 5), (G.N, 5, 1), (G.N, 4, 3), (G.N, 1, 6), (G.N, 0), (G.N, 0, 3)]:
        x = rng.standard_normal(shape)
        for src in (x, np.asfortranarray(x), x.astype(np.float32)):
            d = G.to_device(src)
            assert isinstance(d, engine.DeviceArray) and d.shape == shape and d.ndim == len(shape)
            back = np.asarray(d)
            assert back.dtype == np.float64 and back.shape == shape
            assert rel_err(back, np.asarray(src, dtype=np.float64)) <= max(tol, 1e-7 if src.dtype == np.float32 else tol)
            d.free()
    ints = rng.integers(-4, 5, size=(G.N, 3))
    assert np.array_equal(np.asarray(G.to_device(ints)), ints.astype(float))
    with pytest.raises(ValueError):
        G.to_device(np.zeros(G.N + 1))
    with pytest.raises(TypeError):
        G.to_device(np.zeros(G.N) * 1j)
    d = G.to_device(np.zeros(G.N))
    d.free()
    with pytest.raises(ValueError):
        np.asarray(d)


def test_doctest_chain_stays_on_the_device(golden_doctest):
    """filter.py:232-256 (heat -> MexicanHat analysis -> synthesis, 0.27649) with one upload and one download:
    every intermediate is a DeviceArray with the reference's shape, and the bits equal the host-array chain."""
    g = golden_doctest
    G = graphs.Graph(csr_from(g, "W"))
    G._lmax = float(g["lmax"])
    delta = np.zeros(G.N)
    delta[13] = 1
    heat, mh = filters.Heat(G, 3), filters.MexicanHat(G, Nf=4)
    h1 = heat.filter(delta)
    h2 = mh.analyze(h1)
    h3 = mh.synthesize(h2)
    d1 = heat.filter(G.to_device(delta))
    assert isinstance(d1, engine.DeviceArray) and d1.shape == (G.N,)
    d2 = mh.analyze(d1)
    assert isinstance(d2, engine.DeviceArray) and d2.shape == (G.N, 4)
    d3 = mh.synthesize(d2)
    assert isinstance(d3, engine.DeviceArray) and d3.shape == (G.N,)
    assert np.array_equal(np.asarray(d1), h1) and np.array_equal(np.asarray(d2), h2)
    assert np.array_equal(np.asarray(d3), h3)
    assert "{:.5f}".format(np.linalg.norm(np.asarray(d1) - np.asarray(d3))) == "0.27649"
    assert rel_err(np.asarray(d1), g["s1"]) < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_array_shape_rules_equal_the_host_path(dtype):
    """The truth table of Filter.filter on DeviceArrays: same result shapes, same errors and the same
    numbers as the numpy-in / numpy-out call - including the layouts that need a repack on the device (an
    (N, Nf) panel of Nf signals read as one signal with Nf features, filter.py:270-278)."""
    N = 257
    G = small_graph(N, seed=12, dtype=dtype)
    rng = np.random.default_rng(1)
    heat, mh = filters.Heat(G, 5), filters.MexicanHat(G, Nf=6)
    for shape, out1, out6 in [((N,), (N,), (N, 6)), ((N, 1), (N,), (N, 6)), ((N, 1, 1), (N,), (N, 6)),
                              ((N, 4), (N, 4), (N, 4, 6)), ((N, 4, 1), (N, 4), (N, 4, 6)),
                              ((N, 6), (N, 6), (N,)), ((N, 4, 6), None, (N, 4)), ((N, 1, 6), None, (N,)),
                              ((N, 6, 6), None, (N, 6)), ((N, 6, 1), (N, 6), (N, 6, 6)),
                              ((N, 4, 3), None, None), ((N, 0), (N, 0), (N, 0, 6))]:
        s = rng.standard_normal(shape)
        for bank, expect in ((heat, out1), (mh, out6)):
            d = G.to_device(s)
            if expect is None:
                with pytest.raises(ValueError):
                    bank.filter(d, order=12)
                with pytest.raises(ValueError):
                    bank.filter(s, order=12)
                continue
            y = bank.filter(d, order=12)
            assert isinstance(y, engine.DeviceArray) and y.shape == expect, (shape, bank.Nf)
            assert np.array_equal(np.asarray(y), bank.filter(s, order=12)), (shape, bank.Nf)
    # a result fed back in another reading: 6 heat-filtered signals (N, 6) as ONE signal with 6 features
    six = heat.filter(G.to_device(rng.standard_normal((N, 6))), order=12)
    assert six.cube == (N, 6, 1)
    y = mh.filter(six, order=12)
    assert y.shape == (N,) and np.array_equal(np.asarray(y), mh.filter(np.asarray(six), order=12))
    # ... and an analysis result (N, 6) [1 signal x 6 features, stored as planes] read by a 1-filter bank as 6 signals
    planes = mh.filter(G.to_device(rng.standard_normal(N)), order=12)
    assert planes.cube == (N, 1, 6)
    y = heat.filter(planes, order=12)
    assert y.shape == (N, 6) and np.array_equal(np.asarray(y), heat.filter(np.asarray(planes), order=12))
    with pytest.raises(ValueError):
        mh.analyze(G.to_device(rng.standard_normal((N, 4, 6))))
    with pytest.raises(ValueError):
        mh.synthesize(G.to_device(rng.standard_normal((N, 4))))
    with pytest.raises(TypeError):
        heat.filter(G.to_device(rng.standard_normal(N)), order=0)
    with pytest.raises(ValueError):
        heat.filter(G.to_device(rng.standard_normal(N)), method="lanczos")
    # a DeviceArray of another graph's context / dtype is refused, not reinterpreted
    other = graphs.Graph(random_graph(N, 6, 3), compute_dtype=np.float32 if dtype == np.float64 else np.float64)
    other.estimate_lmax("bounds")
    with pytest.raises(ValueError):
        filters.Heat(other, 5).filter(G.to_device(rng.standard_normal(N)))


def test_chain_at_size_against_the_oracle():
    """heat -> bank analysis -> synthesis on 50k vertices x 16 signals, device-resident, against the oracle's
    chain (one upload, one download); fp64 bar 1e-5 (met at 1e-11)."""
    G = graphs.Sensor(50000, k=6, seed=1)
    G.estimate_lmax("bounds")
    x = np.random.default_rng(2).standard_normal((G.N, 16))
    heat, mh = filters.Heat(G, 10), filters.MexicanHat(G, Nf=4)
    d = mh.synthesize(mh.analyze(heat.filter(G.to_device(x), order=20), order=20), order=20)
    assert d.shape == (G.N, 16)
    L = orc.laplacian(G.W)
    r1 = orc.filter_chebyshev(L, G.lmax, [orc.heat_kernel(10, G.lmax)], x, 20)
    kern = orc.mexican_hat_kernels(G.lmax, 4)
    ref = orc.filter_chebyshev(L, G.lmax, kern, orc.filter_chebyshev(L, G.lmax, kern, r1, 20), 20)
    assert rel_err(np.asarray(d), ref) < 1e-11


# ---- the wrapped Filter.filter of a pygsp-shaped package -------------------------------------------------
def pygsp_like(W, lmax, lap_type="combinatorial"):
    """A module tree with the names plugin.install() patches and a Filter whose filter() / compute_frame() follow
    the reference's control flow (filter.py:303-322, 599-600: analysis = one cheby_op, synthesis = a loop of Nf
    cheby_op calls looked up on the module at call time, the frame = filter(identity)).  Test infrastructure:
    the arithmetic behind the unpatched names is the oracle."""
    mod = types.ModuleType("pygsp")
    mod.filters = types.ModuleType("pygsp.filters")
    approx = mod.filters.approximations = types.ModuleType("pygsp.filters.approximations")
    mod.calls = {"cheby_op": 0, "identity": 0}

    class RefGraph:
        def __init__(self):
            self.W, self.N, self.lap_type = W, W.shape[0], lap_type
            self.L, self.lmax = orc.laplacian(W, lap_type), lmax

        def is_directed(self):
            return False

        def _check_signal(self, s):
            s = np.asanyarray(s)
            if s.shape[0] != self.N:
                raise ValueError("First dimension must be the number of vertices G.N = {}, got {}.".format(self.N, s.shape))
            return s

    def reference_cheby_op(G, c, s, **kw):
        mod.calls["cheby_op"] += 1
        return orc.cheby_op(G.L, G.lmax, c, s)

    def compute_cheby_coeff(f, m=30, N=None, i=None):
        cs = [orc.compute_cheby_coeff(k, f.G.lmax, m) for k in f._kernels]
        return cs[0] if len(cs) == 1 else cs

    class Filter:
        def __init__(self, G, kernels):
            self.G, self._kernels, self.Nf = G, list(kernels), len(kernels)

        def filter(self, s, method="chebyshev", order=30):
            if method != "chebyshev":
                raise ValueError("Unknown method {}.".format(method))
            cube = filters._cube_shape(self.G, self.Nf, np.shape(s))
            s = np.asarray(s, dtype=np.float64).reshape(cube)
            c = approx.compute_cheby_coeff(self, m=order)
            if cube[2] == 1:
                y = approx.cheby_op(self.G, c, s[:, :, 0])
                out = y.reshape((self.G.N, self.Nf, cube[1]), order="F").swapaxes(1, 2)
            else:
                out = np.zeros((self.G.N, cube[1]))
                for i in range(self.Nf):
                    out += approx.cheby_op(self.G, c[i], s[:, :, i])
            return out.squeeze()

        def compute_frame(self, **kwargs):
            mod.calls["identity"] += 1
            return self.filter(np.identity(self.G.N), **kwargs).T.reshape(-1, self.G.N)

    approx.cheby_op = mod.filters.cheby_op = reference_cheby_op
    approx.compute_cheby_coeff = compute_cheby_coeff
    mod.filters.Filter = Filter
    mod.RefGraph = RefGraph
    return mod


def test_plugin_wraps_filter_synthesis_and_frame():
    """install(wrap_filter=True): a synthesis through the patched ``Filter.filter`` is ONE device call whose bits
    equal the mirror class's fused synthesis (not Nf cheby_op calls); ``compute_frame`` never builds the N x N
    identity; install(wrap_filter=False) / uninstall() put the package's own methods back."""
    from pygsp_amd import plugin

    W = random_graph(700, 6, 31)
    lmax = upper_lmax(W)
    fake = pygsp_like(W, lmax)
    G = fake.RefGraph()
    kernels = orc.mexican_hat_kernels(lmax, 5)
    bank = fake.filters.Filter(G, kernels)
    own_filter, own_frame = fake.filters.Filter.filter, fake.filters.Filter.compute_frame
    rng = np.random.default_rng(6)
    x, cube = rng.standard_normal((G.N, 3)), rng.standard_normal((G.N, 3, 5))
    ref_a, ref_s = bank.filter(x, order=18), bank.filter(cube, order=18)
    assert fake.calls["cheby_op"] == 1 + 5

    mirror_G = graphs.Graph(W)
    mirror_G._lmax = lmax
    mirror = filters.Filter(mirror_G, kernels)
    calls = {"n": 0}
    real_cheby_op = filters.cheby_op

    def counting(*a, **k):
        calls["n"] += 1
        return real_cheby_op(*a, **k)

    try:
        plugin.install(fake)
        assert fake.filters.Filter.filter is not own_filter and fake.filters.Filter.compute_frame is not own_frame
        fake.calls["cheby_op"] = 0
        ya, ys = bank.filter(x, order=18), bank.filter(cube, order=18)
        assert fake.calls["cheby_op"] == 0  # nothing ran on the host
        assert ya.shape == (G.N, 3, 5) and rel_err(ya, ref_a) < 1e-12
        assert ys.shape == (G.N, 3) and rel_err(ys, ref_s) < 1e-12
        assert np.array_equal(ys, mirror.filter(cube, order=18))  # the fused Clenshaw synthesis, bit for bit
        assert np.array_equal(ya, mirror.filter(x, order=18))
        # one product cheby_op call for the analysis, none for the synthesis (one gspx_cheby_filter(mode=1))
        filters.cheby_op, before = counting, calls["n"]
        try:
            bank.filter(cube, order=18)
            assert calls["n"] == before
            bank.filter(x, order=18)
            assert calls["n"] == before + 1
        finally:
            filters.cheby_op = real_cheby_op
        # device-resident arrays through the patched class
        d = bank.filter(bank.filter(plugin.to_device(G, x), order=18), order=18)
        assert isinstance(d, engine.DeviceArray) and d.shape == (G.N, 3)
        assert np.array_equal(np.asarray(d), bank.filter(ya, order=18))
        # frame: device identity panels, the package's compute_frame (and its np.identity) is not used
        F = bank.compute_frame(order=12)
        assert fake.calls["identity"] == 0 and F.shape == (5 * G.N, G.N)
        plugin.install(fake, wrap_filter=False)
        assert fake.filters.Filter.filter is own_filter and fake.filters.Filter.compute_frame is own_frame
        assert rel_err(F, bank.compute_frame(order=12)) < 1e-12 and fake.calls["identity"] == 1
        fake.calls["cheby_op"] = 0
        assert rel_err(bank.filter(cube, order=18), ref_s) < 1e-12  # the package's loop over the patched cheby_op
        with pytest.raises(ValueError):
            bank.filter(x, method="lanczos")
        plugin.install(fake)
        with pytest.raises(ValueError):
            bank.filter(x, method="lanczos")  # not chebyshev: the package's own code answers
    finally:
        plugin.uninstall(fake)
    assert fake.filters.Filter.filter is own_filter and fake.filters.Filter.compute_frame is own_frame
    assert fake.filters.approximations.cheby_op.__name__ == "reference_cheby_op"


def test_plugin_frame_of_20k_vertices_without_a_host_identity(monkeypatch):
    """``Filter.compute_frame`` of a 20,000-vertex graph through the wrapped seam (filter.py:593-600 would build
    np.identity(N): 3.2 GB, and ship it): no N x N array is built on the host, the deltas are written on the
    device 1024 at a time; rows checked against the oracle's filtering of single deltas."""
    from pygsp_amd import plugin

    G0 = graphs.Sensor(20000, k=6, seed=4)
    W = G0.W
    lmax = upper_lmax(W)
    fake = pygsp_like(W, lmax)
    G = fake.RefGraph()
    bank = fake.filters.Filter(G, [orc.heat_kernel(20, lmax)])

    def no_identity(*a, **k):
        raise AssertionError("np.identity was called")

    try:
        plugin.install(fake)
        monkeypatch.setattr(np, "identity", no_identity)
        F = bank.compute_frame(order=10)
    finally:
        monkeypatch.undo()
        plugin.uninstall(fake)
    assert F.shape == (G.N, G.N) and fake.calls["identity"] == 0
    picks = [0, 1023, 1024, 7777, 19999]
    deltas = np.zeros((G.N, len(picks)))
    deltas[picks, np.arange(len(picks))] = 1
    ref = orc.cheby_op(G.L, lmax, orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 10), deltas)
    assert rel_err(F[picks, :].T, ref) < 1e-12
    assert np.isfinite(F).all() and np.count_nonzero(F[0]) < G.N // 4  # a localised kernel, not a dense row


def test_plugin_wrap_in_float32_and_with_a_device_list():
    """The wrapped seam in the fp32 engine (`install(dtype=np.float32)`: 1e-3 bar, DeviceArrays in float32) and with
    a device list (`install(devices=[0, 0])`: analysis and the fused synthesis split their signal columns over two
    contexts; DeviceArrays are refused there - they live on one GPU)."""
    from pygsp_amd import plugin

    W = random_graph(900, 6, 17)
    lmax = upper_lmax(W)
    fake = pygsp_like(W, lmax)
    G = fake.RefGraph()
    bank = fake.filters.Filter(G, orc.mexican_hat_kernels(lmax, 3))
    rng = np.random.default_rng(8)
    x, cube = rng.standard_normal((G.N, 5)), rng.standard_normal((G.N, 5, 3))
    ref_a, ref_s = bank.filter(x, order=14), bank.filter(cube, order=14)
    try:
        plugin.install(fake, dtype=np.float32)
        ya, ys = bank.filter(x, order=14), bank.filter(cube, order=14)
        assert ya.dtype == np.float64 and rel_err(ya, ref_a) < 2e-5 and rel_err(ys, ref_s) < 2e-5
        d = bank.filter(bank.filter(plugin.to_device(G, x), order=14), order=14)
        assert d.dtype == np.float32 and d.shape == (G.N, 5)
        assert rel_err(np.asarray(d), bank.filter(ref_a, order=14)) < 1e-4
        plugin.install(fake, devices=[0, 0])
        assert np.max(np.abs(bank.filter(x, order=14) - ref_a)) < 1e-12 * np.max(np.abs(ref_a))
        assert np.max(np.abs(bank.filter(cube, order=14) - ref_s)) < 1e-12 * np.max(np.abs(ref_s))
        with pytest.raises(ValueError):
            bank.filter(plugin.to_device(G, x), order=14)
    finally:
        plugin.uninstall(fake)
        plugin.install(fake)  # back to the single-device float64 configuration for whatever runs next
        plugin.uninstall(fake)


def test_plugin_auto_evaluation_and_placement_tuning(monkeypatch):
    """The two opt-ins of round 6 through the seam of a pygsp-shaped package: install(evaluation='auto') runs a single
    filter's analysis in product (else Newton) form when the panel is large and the guard clears the polynomial (the
    recurrence otherwise: banks, synthesis, small panels), to rounding of the reference's result; plugin.tune_placement draws
    candidate backings for the work panels with the bank's own coefficients and leaves results bit-identical."""
    from pygsp_amd import plugin

    G0 = graphs.Sensor(60000, k=6, seed=9)
    W = G0.W
    lmax = upper_lmax(W)
    fake = pygsp_like(W, lmax)
    G = fake.RefGraph()
    heat = fake.filters.Filter(G, [orc.heat_kernel(20, lmax)])
    bank = fake.filters.Filter(G, orc.mexican_hat_kernels(lmax, 3))
    x = np.random.default_rng(3).standard_normal((G.N, 64))
    ref = orc.cheby_op(G.L, lmax, orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 30), x[:, :2])
    try:
        plugin.install(fake, evaluation="auto")
        monkeypatch.setattr(filters, "AUTO_MIN_PANEL_BYTES", 1 << 20)  # (60k x 64 fp64 = 30 MB: "large" for this test)
        y_auto = heat.filter(x, order=30)
        c30 = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 30)
        expect = "product" if filters.product_guard(c30, np.float64)[0] else "newton"
        assert G._gspx_last_evaluation == expect and rel_err(y_auto[:, :2], ref) < 1e-11
        for form in ("newton", "product"):  # each form on its own through the seam
            plugin.install(fake, evaluation=form)
            assert rel_err(heat.filter(x, order=30)[:, :2], ref) < 1e-11 and G._gspx_last_evaluation == form
        plugin.install(fake, evaluation="auto")
        bank.filter(x[:, :4], order=20)
        assert G._gspx_last_evaluation == "recurrence"  # a bank keeps the reference's recurrence
        plugin.install(fake)  # evaluation back to the default
        y_rec = heat.filter(x, order=30)
        assert G._gspx_last_evaluation == "recurrence" and rel_err(y_rec[:, :2], ref) < 1e-11
        rep = plugin.tune_placement(heat, 64, order=30, candidates=3, stride_mb=256)
        assert len(rep["launch_ms"]) == 3 and rep["kept"] in (0, 1, 2)
        assert np.array_equal(heat.filter(x, order=30), y_rec)
        with pytest.raises(ValueError, match="single-filter"):
            plugin.tune_placement(bank, 64)
        with pytest.raises(ValueError, match="evaluation must be"):
            plugin.install(fake, evaluation="horner")
    finally:
        monkeypatch.undo()
        plugin.uninstall(fake)
