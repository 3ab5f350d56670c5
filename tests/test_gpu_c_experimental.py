"""The EXPERIMENTAL build of libgspx (make -C pygsp_amd/csrc experimental -> _lib/libgspx_exp.so; load it with
GSPX_LIB_PATH): kernels the default library does not contain because they measured slower (fused Newton pair, the small
pair kernel) or are not cleared to run at size (two orders of the three-term recurrence per launch).  Skipped on the
default library - the default `-m gpu` run never launches them.  Real MI355X (`-m gpu`).

    make -C pygsp_amd/csrc experimental
    GSPX_LIB_PATH=$PWD/pygsp_amd/_lib/libgspx_exp.so python -m pytest tests/test_gpu_c_experimental.py -m gpu
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from gpu_helpers import TOL, random_graph, upper_lmax
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs


def _experimental_library():
    try:
        _capi.load()
    except Exception:
        return False
    return _capi.experimental


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _experimental_library(),
                                                  reason="needs the experimental build (GSPX_LIB_PATH=.../libgspx_exp.so)")]
# k_cheb_pair took the GPU box down twice at full size in round 4 (profiles/r04_pair_experiment.md): its tests need a
# second, explicit opt-in on top of the experimental library
pair_optin = pytest.mark.skipif(os.environ.get("GSPX_PAIR_EXPERIMENT") != "1",
                                reason="two-orders-per-launch kernel: opt-in (GSPX_PAIR_EXPERIMENT=1)")


@pytest.fixture(scope="module")
def ctx():
    from pygsp_amd import experimental
    experimental.attach()
    return engine.default_context(0)


def banded_graph(n, per_row, reach, seed):
    """Ragged symmetric weighted graph whose edges stay within `reach` positions (so that 2-hop closures of a row
    block fit LDS), with a few isolated vertices."""
    from scipy import sparse
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), per_row)
    c = np.clip(r + rng.integers(-reach, reach + 1, r.size), 0, n - 1)
    keep = (r != c) & (rng.uniform(size=r.size) < 0.8)
    A = sparse.coo_matrix((rng.uniform(0.1, 1.0, keep.sum()), (r[keep], c[keep])), shape=(n, n)).tocsr()
    W = sparse.lil_matrix(A + A.T)
    for v in (0, 777, n - 1):
        W[v, :] = 0
        W[:, v] = 0
    W = sparse.csr_matrix(W)
    W.eliminate_zeros()
    W.sum_duplicates()
    W.sort_indices()
    return W


@pair_optin
def test_pair_tiles_of_a_random_graph_are_refused(ctx):
    """No vertex locality: the 2-hop closure of a row block is most of the graph and does not fit LDS - an error,
    never a wrong result."""
    W = random_graph(3000, 5, seed=8, isolated=3)
    dev = engine.DeviceGraph.from_w(W, ctx=ctx)
    dev.enable_cheb_pair_tiles(64)
    bx = ctx.upload(np.zeros((3000, 2)))
    with pytest.raises(ValueError, match="LDS"):
        dev.cheby_pair_filter_dev(np.ones(5), bx.ptr, bx.ptr + 0, 2, upper_lmax(W), chunk_lanes=16)
    bx.free()


@pair_optin
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("block_rows", [64, 128, 256])
@pytest.mark.parametrize("build,grid", [(2, 0), (2, 8), (1, 8)])
def test_pair_kernel_equals_the_oracle(ctx, dtype, block_rows, build, grid):
    """A k-NN graph in curve order (what the tiles are for) and a ragged random graph with isolated vertices:
    every chunk width, panels narrower / wider than a chunk, orders 2 ... 30 (1, 2, odd and even pair counts).
    build: 2 the pipelined kernel of round 5 (k_cheb_pair2; shapes it has no build for run the first one), 1 the
    first kernel; grid 8: eight persistent workgroups, so that every one walks MANY blocks (the shape of the
    full-size runs: profiles/r05_pair_experiment.md rung a), 0: what fits."""
    ctx.set_option("pair_kernel", build)
    ctx.set_option("pair_workgroups", grid)
    tol = TOL[np.dtype(dtype)]
    G = graphs.Sensor(9000, k=7, seed=5, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    cases = [(G.device_graph(), orc.laplacian(G.W), float(G.lmax), G.N)]
    W = banded_graph(3000, 5, 30, seed=8)
    cases.append((engine.DeviceGraph.from_w(W, dtype=dtype, ctx=ctx), orc.laplacian(W), upper_lmax(W), 3000))
    rng = np.random.default_rng(block_rows)
    vec = 16 // np.dtype(dtype).itemsize
    for dev, L, lmax, N in cases:
        st = dev.enable_cheb_pair_tiles(block_rows)
        assert st["nb"] == (N + block_rows - 1) // block_rows and st["max_n2"] >= st["max_n1"] >= min(block_rows, N)
        ran = 0
        for K, nsig, lanes in [(2, vec, 4), (4, 2 * vec, 2), (6, 3 * vec, 4), (8, 8 * vec, 8), (30, 5 * vec, 4),
                               (30, 16 * vec, 16), (12, 9 * vec, 8)]:
            c = orc.compute_cheby_coeff(orc.heat_kernel(7, lmax), lmax, K)
            x = rng.standard_normal((N, nsig)).astype(dtype)
            bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
            try:
                ms = dev.cheby_pair_filter_dev(c, bx.ptr, by.ptr, nsig, lmax, chunk_lanes=lanes)
            except ValueError as e:  # wide chunks of large blocks do not fit a workgroup's LDS: refused, not wrong
                assert "LDS" in str(e) and block_rows * lanes >= 1024, (block_rows, lanes, str(e))
                bx.free()
                by.free()
                continue
            ran += 1
            y = by.download(x.shape, dtype)
            assert ms > 0 and ctx.last_timing()["step_launches"] == K // 2
            ref = orc.cheby_op(L, lmax, c, x.astype(np.float64))
            assert rel_err(y, ref) < tol, (K, nsig, lanes)
            one = np.empty_like(x)
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
            one = by.download(x.shape, dtype)
            assert rel_err(y, one) < tol
            bx.free()
            by.free()
        assert ran >= 4
        with pytest.raises(ValueError):
            dev.cheby_pair_filter_dev(np.ones(4), 0, 0, vec, lmax)  # odd order
        dev.disable_cheb_pair_tiles()
        bx = ctx.upload(np.zeros((N, vec), dtype=dtype))
        with pytest.raises(ValueError):
            dev.cheby_pair_filter_dev(np.ones(5), bx.ptr, bx.ptr, vec, lmax)  # no tiles
        bx.free()
    ctx.set_option("pair_kernel", 2)
    ctx.set_option("pair_workgroups", 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_newton_pair_kernel(ctx, dtype):
    tol = TOL[np.dtype(dtype)] * 10
    rng = np.random.default_rng(21)
    W, coords = graphs.sensor_weights(20000, k=8, seed=9)
    L = orc.laplacian(W)
    lmax = upper_lmax(W)
    for perm in (engine.locality_order(W, coords), None):
        dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=perm, ctx=ctx)
        stats = dev.enable_pair_tiles()
        assert stats["max_n1"] <= stats["max_n2"]
        if perm is not None:  # locality order: (nearly) every block stages its entries in LDS
            assert stats["unstaged_blocks"] * 20 < stats["nb"], stats
        for nsig in (4, 8, 32, 64, 100, 128):
            x = rng.standard_normal((W.shape[0], nsig))
            for order in (30, 7, 2, 1):
                c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, order)
                nodes, d = filters.cheb_to_newton(c)
                ref = orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64))
                ctx.set_option("newton_pair", 1)
                y1, _ = dev.newton_filter(nodes, d, x, lmax)
                ctx.set_option("newton_pair", 0)
                y0, _ = dev.newton_filter(nodes, d, x, lmax)
                ctx.set_option("newton_pair", 1)
                assert rel_err(y0, ref) < tol, (nsig, order, "single")
                assert rel_err(y1, ref) < tol, (nsig, order, "pair")
        dev.disable_pair_tiles()
        dev.destroy()
    # a graph with isolated vertices, a hub and ragged rows
    Wr = random_graph(5000, 7, seed=31, hub=True, isolated=5)
    Lr = orc.laplacian(Wr)
    lm = upper_lmax(Wr)
    dev = engine.DeviceGraph.from_w(Wr, dtype=dtype, perm=engine.locality_order(Wr, None), ctx=ctx)
    st = dev.enable_pair_tiles()
    assert st["unstaged_blocks"] >= 1  # the hub's row is longer than 32 entries
    x = rng.standard_normal((5000, 16))
    c = orc.compute_cheby_coeff(orc.heat_kernel(9, lm), lm, 12)
    nodes, d = filters.cheb_to_newton(c)
    y, _ = dev.newton_filter(nodes, d, x, lm)
    assert rel_err(y, orc.cheby_op(Lr, lm, c, x.astype(dtype).astype(np.float64))) < tol
    dev.destroy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pair_small_two_orders_per_launch(ctx, dtype):
    """Option pair_small: one- / two-signal calls on a cache-resident graph run two recurrence orders per launch
    (k_pair_small: T_k on the block's 1-hop closure in LDS, T_{k+1} on its rows).  Same polynomial as the single-step
    path - oracle parity at the same tolerance - for even orders 2 ... 30, eager and replayed as a hipGraph; odd orders,
    wider panels and filterbanks keep the single-step path."""
    G = graphs.Sensor(40000, k=7, seed=12, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    dev = G.device_graph()
    assert G.tile_stats and G.tile_stats["enabled"]
    L = orc.laplacian(G.W)
    rng = np.random.default_rng(3)
    tol = TOL[np.dtype(dtype)]
    try:
        for nsig in (1, 2):
            if nsig == 2 and dtype == np.float32:
                pass  # (2 fp32 signals = 8-byte rows: covered like the others)
            x = rng.standard_normal((G.N, nsig)).astype(dtype)
            bx, by = ctx.upload(x), ctx.alloc(x.nbytes)
            for K in (2, 4, 6, 30, 7):
                c = orc.compute_cheby_coeff(orc.heat_kernel(20, lmax), lmax, K)
                ref = orc.cheby_op(L, lmax, c, x.astype(np.float64))
                ctx.set_option("pair_small", 0)
                dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
                y0 = by.download(x.shape, dtype)
                ctx.set_option("pair_small", 1)
                outs = []
                for _ in range(4):  # the second identical call is recorded, the third and fourth replay the graph
                    dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
                    outs.append(by.download(x.shape, dtype))
                assert rel_err(outs[0], ref) < tol and rel_err(y0, ref) < tol, (nsig, K)
                assert all(np.array_equal(o, outs[0]) for o in outs[1:]), (nsig, K)
                if K == 7:  # odd order: the single-step path either way
                    assert np.array_equal(outs[0], y0)
            bx.free()
            by.free()
        # three signals, and a bank of two filters: not this path (bits equal the default)
        x = rng.standard_normal((G.N, 3)).astype(dtype)
        c2 = np.array([orc.compute_cheby_coeff(orc.heat_kernel(t, lmax), lmax, 8) for t in (5, 9)])
        ctx.set_option("pair_small", 0)
        a3, _ = dev.cheby_filter(c2[0], x, lmax)
        b1, _ = dev.cheby_filter(c2, x[:, :1], lmax)
        ctx.set_option("pair_small", 1)
        assert np.array_equal(dev.cheby_filter(c2[0], x, lmax)[0], a3)
        assert np.array_equal(dev.cheby_filter(c2, x[:, :1], lmax)[0], b1)
    finally:
        ctx.set_option("pair_small", 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_kernel_on_128_row_blocks_is_bit_identical(ctx, dtype):
    """Option tile_br128 (round-5 experiment, profiles/r05_narrow_rows.md): rows of 96 to 128 bytes run the 8-lane build
    of k_step_tile on 128-row blocks (their own row lists, 8-bit positions in tiles of up to 256 rows).  Same row
    products in the same order: the result equals the default's bit for bit, single filters and synthesis steps."""
    G = graphs.Sensor(60000, k=7, seed=21, compute_dtype=dtype)
    G.estimate_lmax("bounds")
    assert G.tile_stats and G.tile_stats["enabled"]
    rng = np.random.default_rng(5)
    vec = 16 // np.dtype(dtype).itemsize
    try:
        for nsig in (6 * vec, 7 * vec, 8 * vec):  # 96-, 112-, 128-byte rows
            x = rng.standard_normal((G.N, nsig)).astype(dtype)
            outs = []
            for setting in (0, 1):
                ctx.set_option("tile_br128", setting)
                y = filters.Heat(G, 10).filter(x, order=21)
                bank = filters.MexicanHat(G, Nf=3)
                z = bank.synthesize(bank.analyze(x[:, :nsig], order=12), order=12)
                outs.append((y, z))
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), nsig
            ref = orc.filter_chebyshev(orc.laplacian(G.W), G.lmax, [orc.heat_kernel(10, G.lmax)], x.astype(np.float64), 21)
            assert rel_err(outs[1][0], ref) < TOL[np.dtype(dtype)]
    finally:
        ctx.set_option("tile_br128", 0)
