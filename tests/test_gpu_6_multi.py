"""the path's one collective.  gspx_gather across contexts in one process, and (below)
two ranks under torch.distributed sharing this box's GPU.  `-m gpu`."""
import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu


def test_gather_and_batch_across_contexts(ctx):
    """gspx_gather: buffers of several contexts concatenated on the root (single-process form of the
    final gather; two contexts on this box's one GPU), and engine.filter_batch on top of it."""
    ctx2 = engine.Context(ctx.device)
    try:
        rng = np.random.default_rng(31)
        a = rng.standard_normal(1000)
        b = rng.standard_normal(333).astype(np.float32)
        pa, pb, pc = ctx.upload(a), ctx2.upload(b), ctx2.alloc(0)
        root = ctx.alloc(a.nbytes + b.nbytes + 16)
        engine.gather([pa, pc, pb], root)
        flat = root.download((a.nbytes + b.nbytes + 16,), np.uint8)
        assert np.array_equal(flat[:a.nbytes].view(np.float64), a)
        assert np.array_equal(flat[a.nbytes:a.nbytes + b.nbytes].view(np.float32), b)
        small = ctx.alloc(8)
        with pytest.raises(ValueError):
            engine.gather([pa, pb], small)
        with pytest.raises(ValueError):
            engine.gather([root], root)
        jobs, refs = [], []
        for i, c_ in enumerate((ctx, ctx2, ctx)):
            W, coords = graphs.sensor_weights(3000 + 500 * i, k=6, seed=40 + i)
            lmax = upper_lmax(W)
            dev = engine.DeviceGraph.from_w(W, dtype=np.float64, perm=engine.locality_order(W, coords), ctx=c_)
            c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, 12)
            x = rng.standard_normal((W.shape[0], 5 + i))
            jobs.append((dev, c, x, lmax))
            refs.append(orc.cheby_op(orc.laplacian(W), lmax, c, x))
        outs = engine.filter_batch(jobs, root_ctx=ctx)
        for o, r in zip(outs, refs):
            assert o.shape == (1,) + r.shape and rel_err(o[0], r) < 1e-12
        for d, _, _, _ in jobs:
            d.destroy()
        for buf in (pa, pb, pc, root, small):  # a context must outlive its buffers
            buf.free()
    finally:
        ctx2.close()
