"""Row a6 in depth: every kernel variant, panel width, schedule (fused flush / deferred combine),
batching, hipGraph replay, in-place input, hubs and tiny graphs of cheby_op (approximations.py:58-114)
against the oracle; the plugin seam on a pygsp-like module.  Real MI355X (`-m gpu`)."""
import numpy as np
import pytest
from scipy import sparse

from conftest import rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu

NSIGS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 31, 32, 33, 64, 65, 100, 128, 130, 257]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nsig", NSIGS)
def test_random_graph_vs_oracle(ctx, dtype, nsig):
    n = 3001
    W = random_graph(n, 9, seed=nsig, hub=True, isolated=5)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    rng = np.random.default_rng(100 + nsig)
    x = rng.standard_normal((n, nsig))
    perm = rng.permutation(n).astype(np.int32) if nsig % 2 else None
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
    c1 = orc.compute_cheby_coeff(orc.heat_kernel(7, lmax), lmax, 25)
    y, ms = dev.cheby_filter(c1, x, lmax)
    ref = orc.cheby_op(L, lmax, c1, x.astype(dtype).astype(np.float64))
    assert y.shape == (1, n, nsig) and ms > 0
    assert rel_err(y[0], ref) < TOL[np.dtype(dtype)]
    # a 3-filter bank (deferred combine) and its fused-flush twin
    c3 = np.array([orc.compute_cheby_coeff(k, lmax, 12) for k in orc.mexican_hat_kernels(lmax, 3)])
    ref3 = orc.cheby_op(L, lmax, c3, x.astype(dtype).astype(np.float64)).reshape(3, n, nsig)
    for combine in (0, 1, 2):
        ctx.set_option("combine", combine)
        y3, _ = dev.cheby_filter(c3, x, lmax)
        assert rel_err(y3, ref3) < TOL[np.dtype(dtype)], combine
    ctx.set_option("combine", 0)
    dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kernel_variants_agree(ctx, dtype):
    """Every tuning knob computes the same thing."""
    n, nsig = 5000, 64
    W = random_graph(n, 10, seed=5, hub=True, isolated=3)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    x = np.random.default_rng(1).standard_normal((n, nsig))
    c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 30)
    ref = orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64))
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, ctx=ctx)
    tol = TOL[np.dtype(dtype)]
    try:
        for kern in (1, 5):  # lane-group panel (scalar metadata), LDS-staged CSR slice
            ctx.set_option("kernel", kern)
            for vec in (0, 1, 2, 4):
                for rpw in (1, 2, 4, 32):
                    for remap in (0, 1):
                        ctx.set_option("vec", vec)
                        ctx.set_option("rows_per_wave", rpw)
                        ctx.set_option("xcd_remap", remap)
                        y, _ = dev.cheby_filter(c, x, lmax)
                        assert rel_err(y[0], ref) < tol, (kern, vec, rpw, remap)
        ctx.set_option("kernel", 0)
        ctx.set_option("vec", 0)
        ctx.set_option("rows_per_wave", 0)
        ctx.set_option("xcd_remap", 1)
        ctx.set_option("kernel", 2)  # narrow kernel forced on a wide panel
        for g in (0, 2, 3):
            ctx.set_option("narrow_g_log2", g)
            y, _ = dev.cheby_filter(c, x[:, :16], lmax)
            assert rel_err(y[0], ref[:, :16]) < tol, g
        ctx.set_option("kernel", 0)
        ctx.set_option("narrow_g_log2", -1)
        # signal batching: 64 signals in batches of 24 (24 + 24 + 16)
        ctx.set_option("max_batch", 24)
        y, _ = dev.cheby_filter(c, x, lmax)
        assert rel_err(y[0], ref) < tol
        c3 = np.array([orc.compute_cheby_coeff(k, lmax, 9) for k in orc.mexican_hat_kernels(lmax, 3)])
        ref3 = orc.cheby_op(L, lmax, c3, x.astype(dtype).astype(np.float64)).reshape(3, n, nsig)
        y3, _ = dev.cheby_filter(c3, x, lmax)
        assert rel_err(y3, ref3) < tol
    finally:
        for key, val in (("kernel", 0), ("vec", 0), ("rows_per_wave", 0), ("xcd_remap", 1),
                         ("max_batch", 0), ("narrow_g_log2", -1)):
            ctx.set_option(key, val)
        dev.destroy()


def test_device_resident_chain_and_timing(ctx):
    """Device pointers in/out: no host round trip between two filters; HIP-event timings."""
    G = graphs.Sensor(20000, seed=1)
    G.estimate_lmax("bounds")
    dev = G.device_graph()
    lmax = G.lmax
    c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, 30)
    x = np.random.default_rng(0).standard_normal((G.N, 8))
    bx, by, bz = ctx.upload(x), ctx.alloc(x.nbytes), ctx.alloc(x.nbytes)
    ms1 = dev.cheby_filter_dev(c, bx.ptr, by.ptr, 8, lmax)
    ms2 = dev.cheby_filter_dev(c, by.ptr, bz.ptr, 8, lmax)
    t = ctx.last_timing()
    assert ms1 > 0 and ms2 > 0 and t["step_launches"] == 30
    assert 0 < t["steps_ms"] <= t["total_ms"] * 1.001
    z = bz.download(x.shape, np.float64)
    L = orc.laplacian(G.W)
    ref = orc.cheby_op(L, lmax, c, orc.cheby_op(L, lmax, c, x))
    assert rel_err(z, ref) < 1e-11


def test_plugin_patches_a_pygsp_like_module(ctx):
    """plugin.install(): the reference's two lookup sites get the device cheby_op
    (filters/filter.py:309 via approximations, filters/__init__.py:115 alias)."""
    import types

    from pygsp_amd import plugin

    W = random_graph(300, 6, 21)

    class RefGraph:  # the attributes the path reads from a reference Graph
        def __init__(self):
            self.W = W
            self.N = W.shape[0]
            self.lap_type = "combinatorial"
            self.L = orc.laplacian(W)
            self.lmax = upper_lmax(W)

        def is_directed(self):
            return False

    fake = types.ModuleType("pygsp")
    fake.filters = types.ModuleType("pygsp.filters")
    fake.filters.approximations = types.ModuleType("pygsp.filters.approximations")
    orig = lambda G, c, s, **kw: orc.cheby_op(G.L, G.lmax, c, s)  # noqa: E731
    fake.filters.approximations.cheby_op = orig
    fake.filters.cheby_op = orig
    G = RefGraph()
    c = orc.compute_cheby_coeff(orc.heat_kernel(4, G.lmax), G.lmax, 15)
    s = np.random.default_rng(3).standard_normal((G.N, 2))
    for mode in ("device", "host"):
        plugin.install(fake, laplacian=mode)
        assert fake.filters.approximations.cheby_op is filters.cheby_op
        assert fake.filters.cheby_op is filters.cheby_op
        y = fake.filters.approximations.cheby_op(G, c, s)
        assert y.shape == (G.N, 2) and rel_err(y, orig(G, c, s)) < 1e-12
        assert fake.filters.cheby_op(G, c, s[:, 0]).shape == (G.N,)
        plugin.uninstall(fake)
        assert fake.filters.approximations.cheby_op is orig and fake.filters.cheby_op is orig


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_synthesis_clenshaw_equals_per_filter_loop(ctx, dtype):
    """Synthesis by one vector-coefficient Clenshaw recurrence (K products) == the reference's
    per-filter loop (K*Nf products) == the oracle, for several panel widths and orders."""
    n = 4001
    W = random_graph(n, 8, seed=77, hub=True, isolated=2)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    tol = TOL[np.dtype(dtype)] * 10
    rng = np.random.default_rng(5)
    perm = rng.permutation(n).astype(np.int32)
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
    try:
        for nf, order, nsig in ((3, 12, 1), (6, 30, 5), (2, 1, 16), (4, 2, 33), (5, 20, 64)):
            c = np.array([orc.compute_cheby_coeff(k, lmax, order) for k in orc.mexican_hat_kernels(lmax, nf)])
            s = rng.standard_normal((nf, n, nsig))
            s64 = s.astype(dtype).astype(np.float64)
            ref = sum(orc.cheby_op(L, lmax, c[f], s64[f]) for f in range(nf))
            for mode in (0, 1):
                ctx.set_option("synthesis", mode)
                y, _ = dev.cheby_filter(c, s, lmax, _capi.SYNTHESIS)
                assert y.shape == (n, nsig)
                assert rel_err(y, ref) < tol, (nf, order, nsig, mode)
    finally:
        ctx.set_option("synthesis", 0)
        dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tiny_graphs_and_huge_hub(ctx, dtype):
    tol = TOL[np.dtype(dtype)] * 10
    rng = np.random.default_rng(11)
    # N = 1, 2, 3 (including an isolated vertex and a lone self loop)
    for W in (np.zeros((1, 1)), np.array([[0.0, 2.0], [2.0, 0.0]]),
              np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.5]])):
        n = W.shape[0]
        L = orc.laplacian(W)
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, ctx=ctx)
        for nsig in (1, 5, 64):
            x = rng.standard_normal((n, nsig))
            c = orc.compute_cheby_coeff(orc.heat_kernel(2, 5.0), 5.0, 7)
            y, _ = dev.cheby_filter(c, x, 5.0)
            ref = orc.cheby_op(L, 5.0, c, x.astype(dtype).astype(np.float64))
            assert np.max(np.abs(y[0] - ref)) < tol * max(1.0, np.max(np.abs(ref)))
        dev.destroy()
    # a hub with 3000 neighbours: rows far longer than the LDS slice / the prefetched chunks
    n = 6000
    r = np.zeros(3000, dtype=np.int64)
    c_ = rng.choice(n - 1, size=3000, replace=False) + 1
    hub = sparse.coo_matrix((rng.uniform(0.1, 1, 3000), (r, c_)), shape=(n, n)).tocsr()
    W = random_graph(n, 6, seed=12) + hub + hub.T
    W = sparse.csr_matrix(W)
    W.sum_duplicates()
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    cc = orc.compute_cheby_coeff(orc.heat_kernel(30, lmax), lmax, 20)
    nodes, d = filters.cheb_to_newton(cc)
    for perm in (None, rng.permutation(n).astype(np.int32)):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        for nsig in (1, 16, 64):
            x = rng.standard_normal((n, nsig))
            ref = orc.cheby_op(L, lmax, cc, x.astype(dtype).astype(np.float64))
            for kern in (0, 1, 5) if nsig > 4 else (0,):
                ctx.set_option("kernel", kern)
                y, _ = dev.cheby_filter(cc, x, lmax)
                assert rel_err(y[0], ref) < tol, (nsig, kern)
                y, _ = dev.newton_filter(nodes, d, x, lmax)
                assert rel_err(y, ref) < tol, (nsig, kern, "newton")
            ctx.set_option("kernel", 0)
        dev.destroy()


def test_options_api(ctx):
    assert ctx.get_option("xcd_remap") == 1 and ctx.get_option("alternate_sweep") == 1
    ctx.set_option("rows_per_wave", 8)
    assert ctx.get_option("rows_per_wave") == 8
    ctx.set_option("rows_per_wave", 0)
    for key, bad in (("vec", 3), ("waves_per_block", 5), ("rows_per_wave", -1), ("narrow_g_log2", 9), ("kernel", 3),
                     ("kernel", 4), ("kernel", 6)):
        with pytest.raises(ValueError):
            ctx.set_option(key, bad)
    with pytest.raises(ValueError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        ctx.get_option("no_such_option")
    assert ctx.bench_copy(64 << 20, 3) > 100.0  # GB/s: sanity of the calibration kernels
    assert ctx.bench_read(8 << 20, 20) > 100.0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_gather_kernel(ctx, dtype):
    tol = TOL[np.dtype(dtype)] * 10
    rng = np.random.default_rng(23)
    W, coords = graphs.sensor_weights(20000, k=8, seed=9)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    for perm in (engine.locality_order(W, coords), None):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        stats = dev.enable_gather_tiles()
        if perm is not None:  # locality order: (nearly) every block stages its tile in LDS
            assert stats["slow_blocks"] * 20 < stats["nb"], stats
        else:                 # vertex order of the generator: no locality, the plain-gather path
            assert stats["slow_blocks"] > stats["nb"] // 2, stats
        # (1, 3, 5, 6, 10, 18: rows that are not made of 16-byte pieces take the tile kernels with padded rows;
        # 2 / 4 / 8 / 16: the 2- / 4- / 8-lane builds)
        ctx.set_option("tile_pad", 2)  # (a graph of this size would leave rows under 16 bytes to the sub-wave kernel)
        for nsig in (1, 2, 3, 4, 5, 6, 8, 10, 16, 18, 32, 64, 100, 128):
            x = rng.standard_normal((W.shape[0], nsig))
            x64 = x.astype(dtype).astype(np.float64)
            for order in (30, 7, 3, 2, 1):
                c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, order)
                ref = orc.cheby_op(L, lmax, c, x64)
                ctx.set_option("tile_gather", 1)
                y1, _ = dev.cheby_filter(c, x, lmax)
                ctx.set_option("tile_gather", 0)
                y0, _ = dev.cheby_filter(c, x, lmax)
                ctx.set_option("tile_gather", 1)
                assert rel_err(y0[0], ref) < tol, (nsig, order, "plain")
                assert rel_err(y1[0], ref) < tol, (nsig, order, "tile")
                if order in (30, 2):
                    nodes, d = filters.cheb_to_newton(c)
                    yn, _ = dev.newton_filter(nodes, d, x, lmax)
                    assert rel_err(yn, ref) < tol, (nsig, order, "newton on tiles")
        ctx.set_option("tile_pad", 1)
        # the same tiles built on the device (per-block sort / unique in LDS) instead of numpy
        st_dev = dev.build_gather_tiles()
        assert st_dev["slow_blocks"] == stats["slow_blocks"] and st_dev["nb"] == stats["nb"]
        if perm is not None:
            assert abs(st_dev["mean_n1"] - stats["mean_n1"]) < 1e-9
        c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 9)
        x = rng.standard_normal((W.shape[0], 12))
        yd, _ = dev.cheby_filter(c, x, lmax)
        assert rel_err(yd[0], orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64))) < tol
        # filterbank (deferred combine): its recurrence steps run on the tile kernel too; synthesis does not
        cb = np.stack([orc.compute_cheby_coeff(k, lmax, 12) for k in orc.mexican_hat_kernels(lmax, 3)])
        x = rng.standard_normal((W.shape[0], 8))
        yb, _ = dev.cheby_filter(cb, x, lmax)
        refb = orc.cheby_op(L, lmax, cb, x.astype(dtype).astype(np.float64)).reshape(3, -1, 8)
        assert rel_err(yb, refb) < tol
        for w in (1, 3, 7):  # padded rows through the deferred combine
            yw, _ = dev.cheby_filter(cb, np.ascontiguousarray(x[:, :w]), lmax)
            assert rel_err(yw, refb[:, :, :w]) < tol, w
            if perm is not None:  # (same kernels, same order of operations: the padded columns change nothing)
                ctx.set_option("tile_pad", 0)
                y_plain, _ = dev.cheby_filter(cb, np.ascontiguousarray(x[:, :w]), lmax)
                ctx.set_option("tile_pad", 1)
                assert rel_err(y_plain, refb[:, :, :w]) < tol, w
        # synthesis (vector-coefficient Clenshaw): the extra input panels are summed in the tile kernel
        s3 = rng.standard_normal((3, W.shape[0], 8))
        refs = sum(orc.cheby_op(L, lmax, cb[f], s3[f].astype(dtype).astype(np.float64)) for f in range(3))
        ys, _ = dev.cheby_filter(cb, s3, lmax, mode=_capi.SYNTHESIS)
        assert rel_err(ys, refs) < tol
        for w in (1, 3, 5):  # ... with padded rows
            ctx.set_option("tile_pad", 2)
            yw, _ = dev.cheby_filter(cb, np.ascontiguousarray(s3[:, :, :w]), lmax, mode=_capi.SYNTHESIS)
            ctx.set_option("tile_pad", 1)
            assert rel_err(yw, refs[:, :w]) < tol, w
        dev.disable_gather_tiles()
        dev.destroy()
    # a graph with isolated vertices, a hub and ragged rows; normalized Laplacian
    Wr = random_graph(5000, 7, seed=33, hub=True, isolated=5)
    for lt in ("combinatorial", "normalized"):
        Lr = orc.laplacian(Wr, lt)
        lm = upper_lmax(Wr) if lt == "combinatorial" else 2.0
        dev = engine.DeviceGraph.from_w(Wr, lt, dtype=dtype, perm=engine.locality_order(Wr, None), ctx=ctx)
        st = dev.enable_gather_tiles()
        assert st["slow_blocks"] >= 1  # the hub's block does not fit
        assert dev.build_gather_tiles()["slow_blocks"] == st["slow_blocks"]
        x = rng.standard_normal((5000, 16))
        c = orc.compute_cheby_coeff(orc.heat_kernel(9, lm), lm, 12)
        y, _ = dev.cheby_filter(c, x, lm)
        assert rel_err(y[0], orc.cheby_op(Lr, lm, c, x.astype(dtype).astype(np.float64))) < tol
        dev.destroy()


def test_graph_replay(ctx):
    rng = np.random.default_rng(77)
    W, coords = graphs.sensor_weights(30000, k=6, seed=12)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    dev = engine.DeviceGraph.from_w(W, perm=engine.locality_order(W, coords), ctx=ctx)
    c1 = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, 20)
    c2 = orc.compute_cheby_coeff(orc.heat_kernel(3, lmax), lmax, 20)
    cb = np.stack([orc.compute_cheby_coeff(k, lmax, 15) for k in orc.mexican_hat_kernels(lmax, 3)])
    for nsig in (1, 8, 16):
        if nsig == 16:  # the recorded launches are those of the LDS-staged step kernel
            assert dev.enable_gather_tiles()["slow_blocks"] == 0
        x = rng.standard_normal((30000, nsig))
        bx, by = ctx.upload(x), ctx.alloc(3 * x.nbytes)
        ref1, ref2 = orc.cheby_op(L, lmax, c1, x), orc.cheby_op(L, lmax, c2, x)
        refb = orc.cheby_op(L, lmax, cb, x).reshape(3, -1, nsig)
        ctx.set_option("graph_launch", 1)
        for rep in range(5):   # eager, eager (pointers settled), record + replay, replay, replay
            dev.cheby_filter_dev(c1, bx.ptr, by.ptr, nsig, lmax)
            assert rel_err(by.download((30000, nsig), np.float64), ref1.reshape(30000, nsig)) < 1e-11, rep
        t = ctx.last_timing()
        assert t["total_ms"] == t["steps_ms"] > 0  # a replayed call reports one time
        # other coefficients: the recorded graph must not be reused
        dev.cheby_filter_dev(c2, bx.ptr, by.ptr, nsig, lmax)
        assert rel_err(by.download((30000, nsig), np.float64), ref2.reshape(30000, nsig)) < 1e-11
        # new input in the same buffer: a replay reads the buffer, not a snapshot
        for rep in range(4):
            x2 = rng.standard_normal((30000, nsig))
            bx.upload(x2)
            dev.cheby_filter_dev(c2, bx.ptr, by.ptr, nsig, lmax)
            assert rel_err(by.download((30000, nsig), np.float64),
                           orc.cheby_op(L, lmax, c2, x2).reshape(30000, nsig)) < 1e-11, rep
        bx.upload(x)
        # interleaved with calls that rewrite the workspace / weights
        for rep in range(4):
            dev.cheby_filter_dev(c1, bx.ptr, by.ptr, nsig, lmax)
            assert rel_err(by.download((30000, nsig), np.float64), ref1.reshape(30000, nsig)) < 1e-11
            if rep % 2:
                dev.laplacian_apply(x)
            else:
                nodes, d = filters.cheb_to_newton(c2)
                dev.newton_filter(nodes, d, x, lmax)
        # filterbank (deferred combine) replays too
        for rep in range(4):
            dev.cheby_filter_dev(cb, bx.ptr, by.ptr, nsig, lmax)
            assert rel_err(by.download((3, 30000, nsig), np.float64), refb) < 1e-11, rep
        ctx.set_option("graph_launch", 2)
        bx.free()
        by.free()
    dev.destroy()


def test_graph_replay_tells_graphs_apart(ctx):
    """A recorded call must not be replayed for a DIFFERENT graph that happens to receive the addresses
    of a destroyed one (same sizes -> the allocator hands the same blocks out again): the replay
    identity carries the graph's birth number, and gspx_graph_destroy drops the recording."""
    rng = np.random.default_rng(78)
    n, nsig = 20000, 8
    x = rng.standard_normal((n, nsig))
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    ctx.set_option("graph_launch", 1)
    try:
        for seed in (1, 2, 3):  # same N, k: same array sizes, different weights
            W, _ = graphs.sensor_weights(n, k=6, seed=seed)
            lmax = 2.0 * float(np.ravel(W.sum(0)).max()) * 1.01  # a different lmax per graph, too
            dev = engine.DeviceGraph.from_w(W, ctx=ctx)
            c = orc.compute_cheby_coeff(orc.heat_kernel(10, 30.0), 30.0, 12)  # identical coefficients
            ref = orc.cheby_op(orc.laplacian(W), lmax, c, x).reshape(n, nsig)
            for rep in range(4):
                dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
                assert rel_err(by.download((n, nsig), np.float64), ref) < 1e-11, (seed, rep)
            dev.destroy()
    finally:
        ctx.set_option("graph_launch", 2)
        bx.free()
        by.free()


def test_tile_block_walks(ctx):
    """The strided walk of k_step_tile's persistent workgroups visits every block exactly once whatever
    the workgroup count: results are bit-identical, also when most workgroups have no block."""
    rng = np.random.default_rng(29)
    for n in (3000, 150000):  # 47 blocks (fewer than workgroups) / 2344 blocks
        W, coords = graphs.sensor_weights(n, k=8, seed=5)
        lmax = upper_lmax(W)
        dev = engine.DeviceGraph.from_w(W, dtype=np.float64, perm=engine.locality_order(W, coords), ctx=ctx)
        dev.enable_gather_tiles()
        c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, 11)
        for nsig in (16, 64):
            x = rng.standard_normal((n, nsig))
            y0, _ = dev.cheby_filter(c, x, lmax)
            assert rel_err(y0[0], orc.cheby_op(orc.laplacian(W), lmax, c, x)) < 1e-12
            for opts in ({"tile_workgroups": 8}, {"tile_workgroups": 64}, {"tile_workgroups": 128},
                         {"tile_workgroups": 4096}, {"alternate_sweep": 0}):
                try:
                    for k, v in opts.items():
                        ctx.set_option(k, v)
                    y1, _ = dev.cheby_filter(c, x, lmax)
                    assert np.array_equal(y0, y1), (n, nsig, opts)
                finally:
                    ctx.set_option("tile_workgroups", 0)
                    ctx.set_option("alternate_sweep", 1)
        dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_input_panel(ctx, dtype):
    """Steps 1-2 reading the caller's panel in place of the permute-in copy: identical results, also for
    orders 1 and 2, in-place calls (x aliasing y: the copy path) and graphs without an internal order."""
    rng = np.random.default_rng(37)
    W, coords = graphs.sensor_weights(30000, k=8, seed=3)
    lmax = upper_lmax(W)
    for perm in (engine.locality_order(W, coords), None):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        st = dev.enable_gather_tiles()
        for nsig in (8, 64, 96):
            x = rng.standard_normal((W.shape[0], nsig)).astype(dtype)
            bx, by0, by1 = ctx.upload(x), ctx.alloc(x.nbytes), ctx.alloc(x.nbytes)
            for order in (1, 2, 3, 30):
                c = np.atleast_2d(orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, order))
                ctx.set_option("fuse_input", 0)
                dev.cheby_filter_dev(c, bx.ptr, by0.ptr, nsig, lmax)
                ctx.set_option("fuse_input", 1)
                dev.cheby_filter_dev(c, bx.ptr, by1.ptr, nsig, lmax)
                y0, y1 = by0.download(x.shape, dtype), by1.download(x.shape, dtype)
                if st["slow_blocks"] == 0:
                    assert np.array_equal(y0, y1), (nsig, order)
                ref = orc.cheby_op(orc.laplacian(W), lmax, c[0], x.astype(np.float64))
                assert rel_err(y1, ref) < TOL[np.dtype(dtype)] * 10
                assert np.array_equal(bx.download(x.shape, dtype), x)  # the input panel is never written
                if order == 30 and x.nbytes <= (32 << 20):  # repeated call: recorded, then replayed as a hipGraph
                    for _ in range(3):
                        dev.cheby_filter_dev(c, bx.ptr, by1.ptr, nsig, lmax)
                        assert np.array_equal(by1.download(x.shape, dtype), y1)
                if order == 3:  # in place
                    bz = ctx.upload(x)
                    dev.cheby_filter_dev(c, bz.ptr, bz.ptr, nsig, lmax)
                    assert np.array_equal(bz.download(x.shape, dtype), y0)
                    bz.free()
            for b in (bx, by0, by1):
                b.free()
        dev.destroy()


def test_compute_frame_in_device_panels():
    """Filter.compute_frame (filter.py:506-600) for N > one panel: the identity is produced on the device
    1024 columns at a time (gspx_identity_panel_dev), never as an N x N host array; equals the oracle's
    filtering of np.identity(N), and p_f(L) is symmetric."""
    G = graphs.Sensor(2500, seed=8)
    G.estimate_lmax("bounds")
    bank = filters.MexicanHat(G, Nf=3)
    F = bank.compute_frame(order=20)
    assert F.shape == (3 * G.N, G.N)
    ref = orc.filter_chebyshev(orc.laplacian(G.W), G.lmax, orc.mexican_hat_kernels(G.lmax, 3), np.identity(G.N), 20)
    assert rel_err(F, ref.T.reshape(-1, G.N)) < 1e-11
    blocks = F.reshape(3, G.N, G.N)
    assert np.max(np.abs(blocks - blocks.transpose(0, 2, 1))) < 1e-12
    # localize(i) is sqrt(N) times column i of every block (filter.py:389-391)
    loc = bank.localize(77, order=20)
    assert rel_err(loc, np.sqrt(G.N) * blocks[:, :, 77].T) < 1e-11


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_host_pipeline_equals_one_shot_call(ctx, dtype):
    """gspx_cheby_filter with host arrays (what Filter.filter of the reference hands over, filter.py:146-328):
    the pipelined form - column batches packed into pinned staging by host threads, H2D / kernels / D2H of
    successive batches overlapping on three streams - returns exactly the bytes of the one-shot form (one copy in,
    the kernels, one copy out), for every batch width (ragged last batch included), thread count, filterbanks
    and synthesis."""
    G = graphs.Sensor(60000, seed=4, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    dev, lmax = G.device_graph(), float(G.lmax)
    rng = np.random.default_rng(6)
    c1 = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 20), m=12))
    c3 = np.array(filters.compute_cheby_coeff(filters.MexicanHat(G, Nf=3), m=9))
    x = rng.standard_normal((G.N, 56)).astype(dtype)
    s3 = rng.standard_normal((3, G.N, 24)).astype(dtype)
    try:
        ctx.set_option("host_pipeline", 0)
        y1, _ = dev.cheby_filter(c1, x, lmax)
        y3, _ = dev.cheby_filter(c3, x, lmax)
        ys, _ = dev.cheby_filter(c3, s3, lmax, _capi.SYNTHESIS)
        assert ctx.last_host_timing() is None
        ref = orc.cheby_op(orc.laplacian(G.W), lmax, c1[0], x[:, :3].astype(np.float64))
        assert rel_err(y1[0][:, :3], ref) < TOL[np.dtype(dtype)] * 10
        ctx.set_option("host_pipeline", 2)
        for batch, threads in ((8, 1), (16, 3), (24, 2), (0, 0), (4, 5)):
            ctx.set_option("host_batch", batch)
            ctx.set_option("host_threads", threads)
            p1, ms = dev.cheby_filter(c1, x, lmax)
            tm = ctx.last_host_timing()
            assert tm is not None and tm["batches"] >= 2 and ms > 0 and tm["kernel_ms"] == pytest.approx(ms)
            if batch:
                assert tm["signals_per_batch"] == batch and tm["batches"] == -(-56 // batch)
            # every batch of at least 32-byte rows runs the kernel family of the one-shot call (same order of
            # operations per column): identical bytes.  16-byte rows (4 fp32 signals) go to the sub-wave kernel,
            # which sums a row's entries in another order: equal to rounding.
            same = (lambda a, b: np.array_equal(a, b)) if batch * np.dtype(dtype).itemsize >= 32 or batch == 0 else \
                (lambda a, b: rel_err(a, b) < TOL[np.dtype(dtype)])
            assert same(p1, y1), (batch, threads)
            assert same(dev.cheby_filter(c3, x, lmax)[0], y3), (batch, threads)
            assert same(dev.cheby_filter(c3, s3, lmax, _capi.SYNTHESIS)[0], ys), (batch, threads)
        # the automatic schedule never leaves a ragged tail of under 32-byte rows (it is merged into the batch
        # before it): identical bytes for every width
        ctx.set_option("host_batch", 0)
        ctx.set_option("host_threads", 0)
        for w in (52, 50, 36):
            xs = np.ascontiguousarray(x[:, :w])
            ctx.set_option("host_pipeline", 0)
            one = dev.cheby_filter(c1, xs, lmax)[0]
            ctx.set_option("host_pipeline", 2)
            assert np.array_equal(dev.cheby_filter(c1, xs, lmax)[0], one), w
            assert ctx.last_host_timing() is not None
        # auto mode: a call of this size (25 MB in + out at fp64) stays on the one-shot path, a large one pipelines
        ctx.set_option("host_pipeline", 1)
        dev.cheby_filter(c1, x[:, :8].copy(), lmax)
        assert ctx.last_host_timing() is None
        big = rng.standard_normal((G.N, 128)).astype(dtype)
        yb, _ = dev.cheby_filter(c1, big, lmax)
        tm = ctx.last_host_timing()
        if big.nbytes * 2 >= (48 << 20):
            assert tm is not None and tm["batches"] >= 4
        ctx.set_option("host_pipeline", 0)
        assert np.array_equal(dev.cheby_filter(c1, big, lmax)[0], yb)
        # an in-place call through the C-ABI (y_host == x_host): the pipeline steps aside (it would overwrite - and
        # pre-fault - columns that are still to be read), the one-shot form reads everything first
        import ctypes
        ctx.set_option("host_pipeline", 2)
        inplace = big.copy()
        ms = ctypes.c_double(0)
        cc = np.ascontiguousarray(c1, dtype=np.float64)
        _capi.check(_capi.load().gspx_cheby_filter(dev._h, lmax, 1, cc.shape[1], _capi.ptr(cc), big.shape[1],
                                                   _capi.ptr(inplace), _capi.ptr(inplace), _capi.ANALYSIS, ctypes.byref(ms)))
        assert ctx.last_host_timing() is None and np.array_equal(inplace, yb[0])
    finally:
        for k, v in (("host_pipeline", 1), ("host_batch", 0), ("host_threads", 0)):
            ctx.set_option(k, v)


def test_plugin_estimate_lmax_seam(ctx):
    """plugin.install(lmax='device'): ``Graph.estimate_lmax`` of a pygsp-shaped module (graph.py:858-931) takes its
    'lanczos' branch to the device - Ritz value x 1.01 within the reference's tolerance of the true lambda_max,
    cached through ``_lmax_method``, 'bounds' untouched, and the original method back after uninstall()."""
    import types

    from scipy.sparse import linalg as splinalg

    from pygsp_amd import plugin

    W, _ = graphs.sensor_weights(6000, k=6, seed=8)
    calls = []

    class RefGraph:
        def __init__(self):
            self.W, self.N, self.lap_type = W, W.shape[0], "combinatorial"
            self.L = orc.laplacian(W)
            self._lmax = self._lmax_method = None

        def is_directed(self):
            return False

        def _get_upper_bound(self):
            return upper_lmax(W)

        def estimate_lmax(self, method="lanczos"):  # stands for the reference's ARPACK / bounds code
            calls.append(method)
            self._lmax_method = method
            self._lmax = self._get_upper_bound()

    fake = types.ModuleType("pygsp")
    fake.filters = types.ModuleType("pygsp.filters")
    fake.filters.approximations = types.ModuleType("pygsp.filters.approximations")
    fake.filters.approximations.cheby_op = fake.filters.cheby_op = lambda G, c, s, **kw: None
    fake.graphs = types.ModuleType("pygsp.graphs")
    fake.graphs.Graph = RefGraph
    original = RefGraph.estimate_lmax
    true = float(splinalg.eigsh(orc.laplacian(W).astype(np.float64), k=1, return_eigenvectors=False)[0])
    try:
        plugin.install(fake, lmax="device")
        G = RefGraph()
        G.estimate_lmax()
        assert calls == [] and G._lmax_method == "lanczos"
        assert true * (1 - 5e-3) * 1.01 <= G._lmax <= true * 1.01 * (1 + 1e-9)
        first = G._lmax
        G.estimate_lmax("lanczos")  # cached
        assert G._lmax == first and calls == []
        G.estimate_lmax("bounds")   # the reference's own code
        assert calls == ["bounds"] and G._lmax == upper_lmax(W)
        plugin.install(fake)        # default: the reference's estimate_lmax again
        assert RefGraph.estimate_lmax is original
        plugin.install(fake, lmax="device")
        assert RefGraph.estimate_lmax is not original
    finally:
        plugin.uninstall(fake)
    assert RefGraph.estimate_lmax is original


def test_library_has_no_retired_options(ctx):
    """The library neither exports the retired kernels' entry points (tests/test_capi.py) nor knows their context
    options (two orders per launch, fused Newton pair, small pair kernel, 128-row blocks: retired in round 6)."""
    for key in ("pair_small", "pair_small_mb", "newton_pair", "pair_workgroups", "pair_workgroups_per_cu", "pair_kernel",
                "tile_br128"):
        with pytest.raises(ValueError, match="unknown option"):
            ctx.set_option(key, 1)
    assert not hasattr(engine.DeviceGraph, "cheby_pair_filter_dev")


@pytest.mark.parametrize("dtype,nsig", [(np.float64, 64), (np.float64, 32), (np.float32, 64), (np.float64, 96)])
def test_mix_ceiling_calibration_runs_beside_the_filter(ctx, dtype, nsig):
    """gspx_bench_step_mix (VERDICT r5 "Next 1"): the filter's launches with the row products removed - a calibration
    whose output means nothing.  It must time the same number of launches, leave the context and the graph usable (the
    next real call is right to rounding, eager and replayed), refuse calls that would not run the wide LDS-staged
    step, and never be reachable through an option."""
    G = graphs.Sensor(20000, seed=3, tiles=True, reorder="morton", compute_dtype=dtype, ctx=ctx)
    G.estimate_lmax("bounds")
    dev = G.device_graph(dtype)
    x = np.random.default_rng(2).standard_normal((G.N, nsig)).astype(dtype)
    c = orc.compute_cheby_coeff(orc.heat_kernel(10, G.lmax), G.lmax, 30)
    ref = orc.cheby_op(orc.laplacian(G.W), G.lmax, c, x.astype(np.float64))
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
    dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, G.lmax)
    real = ctx.last_timing()
    for mode in (1, 2):
        t = dev.bench_step_mix(c, bx.ptr, by.ptr, nsig, G.lmax, mode)
        assert t["step_launches"] == real["step_launches"] == 30 and t["steps_ms"] > 0
    for _ in range(3):  # eager, recorded, replayed
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, G.lmax)
        assert rel_err(by.download((G.N, nsig), dtype), ref) < TOL[np.dtype(dtype)]
    assert np.array_equal(bx.download((G.N, nsig), dtype), x)  # the input is never written
    with pytest.raises(ValueError, match="mode"):
        dev.bench_step_mix(c, bx.ptr, by.ptr, nsig, G.lmax, 3)
    with pytest.raises(ValueError, match="wide"):
        dev.bench_step_mix(c, bx.ptr, by.ptr, 128 // np.dtype(dtype).itemsize, G.lmax, 1)  # 128-byte rows: narrow build
    with pytest.raises(ValueError, match="unknown option"):
        ctx.set_option("calib_mix", 1)
    bx.free()
    by.free()


def test_stream_mix_calibration(ctx):
    """gspx_bench_streams (round 6): n read + m write streams walked together, optionally non-temporal, in place, in the
    step's walk or inside the context's own workspace - a calibration whose only contract is a positive rate below the
    chip's peak and the refusal of bad arguments."""
    fresh = engine.Context(0)
    try:  # bit 4: the streams lie in the context's T workspace - a context that has none yet refuses
        with pytest.raises(ValueError, match="workspace"):
            fresh.bench_streams(64 << 20, 1, 1, 16, 8, 2)
    finally:
        fresh.close()
    for nr, nw, nt in ((1, 0, 0), (0, 1, 0), (1, 1, 0), (3, 1, 0), (4, 2, 3), (1, 1, 4), (3, 2, 4), (2, 1, 12), (3, 1, 8)):
        v = ctx.bench_streams(64 << 20, nr, nw, nt, 8, 2)
        assert 500 < v < 16000, (nr, nw, nt, v)
    for bad in ((0, 0, 0), (5, 0, 0), (1, 3, 0), (1, 1, 32)):
        with pytest.raises(ValueError):
            ctx.bench_streams(64 << 20, *bad)
