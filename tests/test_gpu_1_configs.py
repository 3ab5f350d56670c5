"""BASELINE.json configs 1-4 and the headline workload at (or near) their full sizes: oracle columns plus
size-independent properties (linearity, the constant-signal identity p(L) 1 = p(0) 1, reorder invariance).
Through the C-ABI on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

from conftest import rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import engine, filters, graphs

pytestmark = pytest.mark.gpu


def _constant_signal_gain(c):
    """L 1 = 0  =>  T_k(L~) 1 = (-1)^k 1, so filtering a constant multiplies it by
    0.5 c0 + sum_k (-1)^k c_k."""
    k = np.arange(1, len(c))
    return 0.5 * c[0] + np.sum(((-1.0) ** k) * c[1:])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config1_sensor100k_single_signal(dtype):
    """BASELINE.json configs[1]: Sensor(100000), combinatorial, Heat, K=30, 1 signal."""
    G = graphs.Sensor(100000, seed=42, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    h = filters.Heat(G, scale=50)
    s = np.random.default_rng(0).standard_normal(G.N)
    y = h.filter(s, order=30)
    L = orc.laplacian(G.W)
    ref = orc.filter_chebyshev(L, G.lmax, [orc.heat_kernel(50, G.lmax)], s, 30)
    assert rel_err(y, ref) < BAR[np.dtype(dtype)] * 1e-2
    assert abs(G.L - L).max() < (1e-13 if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_headline_size_properties(ctx, dtype):
    """North-star size: 1M-vertex k=8 sensor graph (~10M stored entries), 64 signals, order 30.
    Oracle on ALL 64 columns in float64 (the reference's arithmetic; 48 s of scipy on one core) and on 8 of them in
    float32; linearity and the constant-signal identity on all 64 in both."""
    N, nsig, order = 1000000, 64, 30
    W, coords = graphs.sensor_weights(N, k=8, seed=42)
    G = graphs.Graph(W, coords=coords, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    lmax = G.lmax
    c = orc.compute_cheby_coeff(orc.heat_kernel(50, lmax), lmax, order)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, nsig)).astype(dtype)
    x[:, 5] = 3.0  # a constant column
    dev = G.device_graph()
    ctx.set_option("host_pipeline", 0)  # the whole 64-signal panel as ONE device batch: the headline kernel shape
    y, ms = dev.cheby_filter(c, x, lmax)
    y = y[0]
    ctx.set_option("host_pipeline", 1)  # ... and the pipelined host path (16-signal batches) gives the same bytes
    yp, _ = dev.cheby_filter(c, x, lmax)
    assert ctx.last_host_timing()["batches"] >= 4 and np.array_equal(yp[0], y)
    del yp
    tol = BAR[np.dtype(dtype)]
    # constant-signal identity
    gain = _constant_signal_gain(c)
    assert np.max(np.abs(y[:, 5] - 3.0 * gain)) < tol * abs(3.0 * gain)
    # the oracle: every column in float64, every ninth in float32
    L = orc.laplacian(W)
    cols = list(range(nsig)) if dtype == np.float64 else list(range(0, nsig, 9))
    ref = orc.cheby_op(L, lmax, c, x[:, cols].astype(np.float64))
    assert rel_err(y[:, cols], ref) < tol * 1e-1
    # linearity: f(2 x_a - x_b) = 2 f(x_a) - f(x_b), columnwise
    x2 = (2 * x[:, :32] - x[:, 32:]).astype(dtype)
    y2, _ = dev.cheby_filter(c, x2, lmax)
    lin = 2 * y[:, :32].astype(np.float64) - y[:, 32:].astype(np.float64)
    assert rel_err(y2[0], lin) < (1e-10 if dtype == np.float64 else 1e-3)
    # vertex order is internal: the device graph without reordering gives the same answer
    dev_plain = engine.DeviceGraph.from_w(W, dtype=dtype, perm=None, ctx=ctx)
    y3, _ = dev_plain.cheby_filter(c, x[:, :4], lmax)
    assert rel_err(y3[0], y[:, :4]) < (1e-11 if dtype == np.float64 else 1e-4)
    dev_plain.destroy()


def test_erdos_renyi_filterbank_fp32():
    """Scaled-down BASELINE.json configs[2]: ER graph (isolated vertices occur), MexicanHat x6,
    K=50, fp32 engine vs fp64 oracle at the 1e-3 bar."""
    N = 200000
    G = graphs.ErdosRenyi(N, p=10.0 / N, seed=0, compute_dtype=np.float32)
    G.estimate_lmax("bounds")
    assert G.W.dtype == np.int64
    mh = filters.MexicanHat(G, Nf=6)
    x = np.random.default_rng(1).standard_normal((N, 16)).astype(np.float32)
    y = mh.filter(x, order=50)
    assert y.shape == (N, 16, 6)
    L = orc.laplacian(G.W.astype(np.float64))
    ref = orc.filter_chebyshev(L, G.lmax, orc.mexican_hat_kernels(G.lmax, 6), x[:, :2].astype(np.float64), 50)
    assert rel_err(y[:, :2, :], ref) < 1e-3 * 1e-1


def test_sbm_normalized_isolated_rule():
    """Scaled-down BASELINE.json configs[3]: SBM, normalized Laplacian, 16 signals."""
    N, k = 100000, 16
    G = graphs.StochasticBlockModel(N, k=k, p=12.0 * k / N, q=4.0 * k / (N * (k - 1)), seed=0,
                                    lap_type="normalized")
    G.estimate_lmax("bounds")
    assert G.lmax == 2
    L = orc.laplacian(G.W.astype(np.float64), "normalized")
    assert abs(G.L - L).max() < 1e-14 and G.L.nnz == L.nnz
    x = np.random.default_rng(2).standard_normal((N, 16))
    y = filters.Heat(G, 10).filter(x, order=30)
    ref = orc.filter_chebyshev(L, 2.0, [orc.heat_kernel(10, 2.0)], x[:, :3], 30)
    assert rel_err(y[:, :3], ref) < 1e-11


def test_config2_full_size_parity(ctx):
    """BASELINE.json configs[2] at FULL size: ErdosRenyi(N=1e6, p=1e-5) (~10M stored entries, isolated vertices
    occur), MexicanHat x 6, order 50, 64 signals, fp32 engine, device resident; two oracle columns (fp64) at a
    tenth of the north star's 1e-3 bar, and the constant-signal identity p_f(L) 1 = p_f(0) 1 on a third."""
    N, nsig, K = 1000000, 64, 50
    G = graphs.ErdosRenyi(N, p=10.0 / N, seed=0, compute_dtype=np.float32)
    G.estimate_lmax("bounds")
    bank = filters.MexicanHat(G, Nf=6)
    c = np.array(filters.compute_cheby_coeff(bank, m=K))
    x = np.random.default_rng(0).standard_normal((N, nsig)).astype(np.float32)
    x[:, 5] = 1.0
    dev = G.device_graph()
    bx, by = ctx.upload(x), ctx.alloc(6 * x.nbytes)
    try:
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, float(G.lmax))
        y = by.download((6, N, nsig), np.float32)
    finally:
        bx.free()
        by.free()
    cols = [0, 63]
    L = orc.laplacian(G.W.astype(np.float64))
    ref = orc.cheby_op(L, float(G.lmax), c, x[:, cols].astype(np.float64)).reshape(6, N, 2)
    assert rel_err(y[:, :, cols], ref) < 1e-3 * 1e-1
    for f in range(6):
        gain = _constant_signal_gain(c[f])
        assert np.max(np.abs(y[f][:, 5] - gain)) < 1e-4


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config3_full_size_parity(ctx, dtype):
    """BASELINE.json configs[3] at FULL size: StochasticBlockModel(N=2e6, k=16, p=9.6e-5, q=2.13e-6), normalized
    Laplacian (isolated-vertex rule included), Heat order 30, 16 signals; two oracle columns at a tenth of the
    north star's bar (1e-5 fp64 / 1e-3 fp32)."""
    N, nsig, K = 2000000, 16, 30
    G = graphs.StochasticBlockModel(N, k=16, p=9.6e-5, q=2.13e-6, seed=0, lap_type="normalized", compute_dtype=dtype)
    G.estimate_lmax("bounds")
    assert G.lmax == 2
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 10), m=K))
    x = np.random.default_rng(0).standard_normal((N, nsig)).astype(dtype)
    dev = G.device_graph()
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    try:
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, 2.0)
        y = by.download((N, nsig), dtype)
    finally:
        bx.free()
        by.free()
    cols = [0, 15]
    L = orc.laplacian(G.W.astype(np.float64), "normalized")
    ref = orc.cheby_op(L, 2.0, c[0], x[:, cols].astype(np.float64))
    assert rel_err(y[:, cols], ref) < (1e-5 if dtype == np.float64 else 1e-3) * 1e-1
    if dtype == np.float64:
        assert abs(G.L - L).max() < 1e-14 and G.L.nnz == L.nnz


def test_panel_beyond_the_2gib_descriptor_window(ctx):
    """A signal panel larger than the 2 GiB window of a buffer descriptor (600k vertices x 500 fp64 signals =
    2.4 GB in, 2.4 GB out) is split into column batches inside the call; every batch must land in its own
    columns of the caller's panels.  Oracle on columns of the first, a middle and the last batch, plus the
    constant-signal identity on a column of every batch (p(L) 1 = p(0) 1 for the combinatorial Laplacian)."""
    N, nsig, order = 600000, 500, 6
    W, coords = graphs.sensor_weights(N, k=6, seed=21)
    G = graphs.Graph(W, coords=coords, compute_dtype=np.float64)
    G.estimate_lmax("bounds")
    lmax = G.lmax
    c = orc.compute_cheby_coeff(orc.heat_kernel(30, lmax), lmax, order)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, nsig))
    const_cols = list(range(7, nsig, 50))
    x[:, const_cols] = 2.0
    assert x.nbytes > (1 << 31)
    dev = G.device_graph()
    ctx.set_option("host_pipeline", 0)  # one device call for the whole 2.4 GB panel: the batching under test is the
    y, _ = dev.cheby_filter(c, x, lmax)  # engine's own split at the 2 GiB descriptor window
    ctx.set_option("host_pipeline", 1)
    assert ctx.last_timing()["step_launches"] >= 2 * order  # at least two column batches
    y = y[0]
    gain = _constant_signal_gain(c)
    assert np.max(np.abs(y[:, const_cols] - 2.0 * gain)) < 1e-12 * abs(2.0 * gain) + 1e-13
    cols = [0, 1, 249, 250, 498, 499]
    ref = orc.cheby_op(orc.laplacian(W), lmax, c, x[:, cols])
    assert rel_err(y[:, cols], ref) < 1e-12
    # untouched by the batching: a column's result does not depend on which batch it travels in
    y2, _ = dev.cheby_filter(c, np.ascontiguousarray(x[:, 240:260]), lmax)
    assert rel_err(y2[0][:, [9, 10]], y[:, [249, 250]]) < 1e-14
