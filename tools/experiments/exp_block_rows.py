import sys; sys.path.insert(0,'.')
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float32,np.float64):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    st=dev.build_gather_tiles(); print(st)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    best=1e9
    for _ in range(4):
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); best=min(best,ctx.last_timing()["steps_ms"]/30)
    print(np.dtype(dtype).name,"ms per order %.4f"%best,flush=True)
    bx.free(); by.free(); dev.destroy()
