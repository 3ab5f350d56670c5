import sys; sys.path.insert(0,'.')  # run from the repo root
# narrow panels on a large local graph: LDS-staged step (8-lane groups) vs the plain / narrow kernels
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float64,np.float32):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    dev.build_gather_tiles()
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    for nsig in (2,4,8,16,32):
        x=np.random.default_rng(0).standard_normal((N,nsig)).astype(dtype)
        bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
        res=[]
        for tg in (0,1):
            ctx.set_option("tile_gather",tg)
            best=1e9
            for _ in range(3):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); best=min(best,ctx.last_timing()["steps_ms"]/30)
            res.append(best)
        print(np.dtype(dtype).name,"nsig",nsig,"row bytes",nsig*x.itemsize,"plain %.4f tile %.4f ms/order"%tuple(res),flush=True)
        bx.free(); by.free()
    ctx.set_option("tile_gather",1)
    dev.destroy()
