import sys, time; sys.path.insert(0,'.')  # run from the repo root
# k_step_tile under Morton vs Hilbert internal order (tile size n1, time per order)
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
orders={"morton":engine.locality_order(W,coords),"hilbert":engine.hilbert_order(coords)}
for dtype in (np.float32,np.float64):
  for name,perm in orders.items():
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    st=dev.enable_gather_tiles()
    best=1e9
    for _ in range(4):
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); best=min(best,ctx.last_timing()["steps_ms"]/30)
    print(np.dtype(dtype).name,name,"mean n1 %.1f max %d slow %d"%(st["mean_n1"],st["max_n1"],st["slow_blocks"]),"ms per order %.4f"%best,flush=True)
    bx.free(); by.free(); dev.destroy()
