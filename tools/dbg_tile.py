import numpy as np, sys
sys.path.insert(0,'.')
from oracle import cheby_oracle as orc
from pygsp_amd import engine, graphs
ctx = engine.default_context(0)
W, coords = graphs.sensor_weights(200000, k=8, seed=3)
L = orc.laplacian(W); lmax = 2.0*float(np.ravel(W.sum(0)).max())
rng = np.random.default_rng(0)
for dtype in (np.float64, np.float32):
    dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=engine.locality_order(W, coords), ctx=ctx)
    st = dev.build_gather_tiles()
    for nsig in (8, 16, 32, 64, 96):
        x = rng.standard_normal((W.shape[0], nsig))
        for order in (1, 2, 3, 4, 7, 30):
            c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, order)
            y, _ = dev.cheby_filter(c, x, lmax)
            ref = orc.cheby_op(L, lmax, c, x[:, :2].astype(dtype).astype(np.float64))
            e = np.abs(y[0][:, :2] - ref)
            yfull = None
            print(np.dtype(dtype).name, 'nsig', nsig, 'order', order, 'err', float(e.max()/np.abs(ref).max()), 'bad rows', int((e.max(axis=1) > 1e-4*np.abs(ref).max()).sum()), flush=True)
    dev.destroy()
