import sys; sys.path.insert(0,'.')  # run from the repo root
# plain gather kernels on mid-width fp64 panels (64-128 byte rows): panel kernel (1) vs LDS-staged (5)
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
ctx.set_option("tile_gather",0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float64,):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    for nsig in (4,8,16,32,64):
        x=np.random.default_rng(0).standard_normal((N,nsig)).astype(dtype)
        bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
        out=[]
        for kern in (0,1,5):
            ctx.set_option("kernel",kern)
            best=1e9
            for _ in range(3):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); best=min(best,ctx.last_timing()["steps_ms"]/30)
            out.append(best)
        print(np.dtype(dtype).name,"nsig",nsig,"auto %.4f kernel1 %.4f kernel5 %.4f ms/order"%tuple(out),flush=True)
        bx.free(); by.free()
    dev.destroy()
ctx.set_option("kernel",0); ctx.set_option("tile_gather",1)
