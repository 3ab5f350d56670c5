"""Randomised differential test of the device path against the oracle: random graphs (ragged, hubs,
isolated vertices, sensor graphs), Laplacian types, dtypes, internal orders (none / curve / random
permutation), gather tiles (host- or device-built), engine options, panel widths, orders, filterbanks,
synthesis, repeated calls (hipGraph replay) and the Newton form.  Needs a real MI355X: `-m gpu`."""
import numpy as np
import pytest

from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs
from gpu_helpers import random_graph, upper_lmax

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_against_oracle(seed):
    cases = 50
    rng = np.random.default_rng(seed)
    ctx = engine.default_context(0)
    try:
        _run(cases, rng, ctx)
    finally:
        for k, v in (("kernel", 0), ("tile_gather", 1), ("graph_launch", 2), ("xcd_remap", 1), ("alternate_sweep", 1),
                     ("combine", 0), ("synthesis", 0), ("max_batch", 0), ("host_pipeline", 1), ("host_batch", 0),
                     ("host_edge", 0), ("host_threads", 0), ("tile_pad", 1), ("tile_min_row", 16), ("tile_lg", 0)):
            ctx.set_option(k, v)


def _run(cases, rng, ctx):
    for case in range(cases):
        dtype = [np.float64, np.float32][rng.integers(2)]
        kind = rng.integers(3)
        if kind == 0:
            N = int(rng.integers(2, 4000))
            W = random_graph(N, float(rng.uniform(1, 12)), seed=int(rng.integers(1 << 30)), hub=bool(rng.integers(2)) and N > 400,
                             isolated=int(rng.integers(0, 4)) if N > 10 else 0)
            coords = None
        else:
            N = int(rng.integers(50, 30000))
            W, coords = graphs.sensor_weights(N, k=int(rng.integers(2, 12)), seed=int(rng.integers(1 << 30)))
        lap = ["combinatorial", "normalized"][rng.integers(2)]
        L = orc.laplacian(W, lap)
        lmax = (upper_lmax(W) if lap == "combinatorial" else 2.0) * float(rng.uniform(1.0, 1.3))
        perm_kind = rng.integers(3)
        perm = None if perm_kind == 0 else (engine.locality_order(W, coords) if perm_kind == 1 else rng.permutation(N).astype(np.int32))
        dev = engine.DeviceGraph.from_w(W, lap, dtype=dtype, perm=perm, ctx=ctx)
        tiles = bool(rng.integers(2))
        if tiles:
            (dev.build_gather_tiles if rng.integers(2) else dev.enable_gather_tiles)()
        opts = {"kernel": int(rng.choice([0, 0, 1, 2, 5, 5])), "tile_gather": int(rng.integers(2)), "graph_launch": int(rng.integers(3)),
                "xcd_remap": int(rng.integers(2)), "alternate_sweep": int(rng.integers(2)), "combine": int(rng.integers(3)),
                "synthesis": int(rng.integers(2)), "max_batch": int(rng.choice([0, 0, 0, 8, 20])),
                # narrow / odd panels: padded rows on the tile kernels (always / by size / never), the 2- / 4- / 8-lane builds
                "tile_pad": int(rng.choice([0, 1, 2, 2])), "tile_min_row": int(rng.choice([16, 16, 32])),
                "tile_lg": int(rng.choice([0, 0, 0, 2, 4, 8])),
                # the host-array calls below: one-shot, or pipelined in column batches of any width / thread count
                # (ragged and 1-column batches included: against the oracle, not bit for bit)
                "host_pipeline": int(rng.choice([0, 1, 2, 2])), "host_batch": int(rng.choice([0, 1, 3, 4, 8, 16, 24])),
                "host_edge": int(rng.choice([0, 0, 2, 8])), "host_threads": int(rng.choice([0, 1, 2, 5]))}
        for k, v in opts.items():
            ctx.set_option(k, v)
        nsig = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 32, 33, 64, 100, 130]))
        order = int(rng.integers(1, 60))
        nf = int(rng.choice([1, 1, 1, 2, 4]))
        kernels = [orc.heat_kernel(float(rng.uniform(1, 60)), lmax)] if nf == 1 else orc.mexican_hat_kernels(lmax, nf)
        c = np.stack([orc.compute_cheby_coeff(k, lmax, order) for k in kernels])
        synth = nf > 1 and bool(rng.integers(2))
        tol = 1e-10 if dtype == np.float64 else 3e-4
        for rep in range(int(rng.integers(1, 4))):  # repeats exercise the hipGraph replay
            if synth:
                x = rng.standard_normal((nf, N, nsig))
                ref = sum(orc.cheby_op(L, lmax, c[f], x[f].astype(dtype).astype(np.float64)) for f in range(nf))
                y, _ = dev.cheby_filter(c, x, lmax, mode=_capi.SYNTHESIS)
            else:
                x = rng.standard_normal((N, nsig))
                ref = orc.cheby_op(L, lmax, c, x.astype(dtype).astype(np.float64)).reshape(nf, N, nsig)
                y, _ = dev.cheby_filter(c, x, lmax)
            den = np.max(np.abs(ref)) or 1.0
            err = float(np.max(np.abs(y - ref)) / den)
            if not err < tol:
                raise AssertionError(("case", case, "rep", rep, dict(
                    N=N, kind=int(kind), lap=lap, dtype=np.dtype(dtype).name, perm=int(perm_kind), tiles=tiles, nsig=nsig,
                    order=order, nf=nf, synth=synth, opts=opts, err=err)))
        if nf == 1 and rng.integers(2):
            nodes, d = filters.cheb_to_newton(c[0])
            x = rng.standard_normal((N, nsig))
            yn, _ = dev.newton_filter(nodes, d, x, lmax)
            ref = orc.cheby_op(L, lmax, c[0], x.astype(dtype).astype(np.float64)).reshape(N, nsig)
            err = float(np.max(np.abs(yn - ref)) / (np.max(np.abs(ref)) or 1.0))
            if not err < tol * 100:
                raise AssertionError(("newton", case, dict(N=N, lap=lap, nsig=nsig, order=order, tiles=tiles, opts=opts, err=err)))
        dev.destroy()


def test_soak_slice_of_the_tile_kernel():
    """A two-minute slice of tools/soak.py: the LDS-staged recurrence kernel against oracle columns over panel
    widths x dtypes x orders x repeats on Hilbert-ordered sensor graphs.  (The kernel's tile-DMA race of round 2
    had passed 120 functional tests: rare events need repetition, so a slice of the soak lives in the suite.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak
    runs, bad = soak.soak_kernel(rounds=3, log=lambda *a: None, **soak.SLICE)
    assert runs == 2 * 2 * 6 * 3 * 3 and bad == 0


def test_soak_slice_of_the_host_pipeline():
    """Pipelined host-pointer calls (threads + three streams) against their one-shot form, bit for bit, over random
    batch widths, thread counts, panel widths, filterbanks and synthesis."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak
    runs, bad = soak.soak_host_pipeline(calls=60, log=lambda *a: None)
    assert runs == 60 and bad == 0
