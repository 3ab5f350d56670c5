import sys; sys.path.insert(0,'.')  # run from the repo root
# Is a working set that fits the 256 MB Infinity Cache faster per row?  (decides whether
# cache-level time skewing over row chunks is worth building)
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
ctx.set_option("newton_pair",0)
for dtype in (np.float64,np.float32):
  for N in (31250,62500,125000,250000,500000,1000000):
    W,coords=graphs.sensor_weights(N,k=8,seed=42)
    perm=engine.locality_order(W,coords)
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    nodes,d=filters.cheb_to_newton(c[0])
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    bn=br=1e9
    for _ in range(5):
        dev.newton_filter_dev(nodes,d,bx.ptr,by.ptr,64,lmax); bn=min(bn,ctx.last_timing()["steps_ms"]/30)
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); br=min(br,ctx.last_timing()["steps_ms"]/30)
    ws=(3*x.nbytes+dev.nnz_internal*(x.itemsize+4))/2**20
    print(np.dtype(dtype).name,"N",N,"working set MB %.0f"%ws,"newton ns/row %.4f"%(bn*1e6/N),"recurrence ns/row %.4f"%(br*1e6/N),flush=True)
    bx.free(); by.free(); dev.destroy()
