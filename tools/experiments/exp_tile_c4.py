import sys; sys.path.insert(0,'.')  # run from the repo root
# config-4 size (Sensor 5e5 x 32 signals: panels of 64 / 128 MB, Infinity-Cache resident): which of
# alternate_sweep / tile_nt help or hurt here?
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
for N,nsig in ((500000,32),(250000,32),(1000000,16)):
    W,coords=graphs.sensor_weights(N,k=6,seed=0)
    perm=engine.locality_order(W,coords)
    for dtype in (np.float64,np.float32):
        dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
        lmax=2.0*float(dev.download_dw().max())
        G=type("G",(),{"lmax":lmax,"e":None})()
        c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
        dev.enable_gather_tiles()
        x=np.random.default_rng(0).standard_normal((N,nsig)).astype(dtype)
        bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
        for rep in range(2):
            res=[]
            for alt,nt in ((0,0),(1,0),(0,5),(1,5),(1,1),(1,4)):
                ctx.set_option("alternate_sweep",alt); ctx.set_option("tile_nt",nt)
                b=1e9
                for _ in range(5):
                    dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); b=min(b,ctx.last_timing()["steps_ms"])
                res.append("alt%d/nt%d %.3f"%(alt,nt,b))
            print(N,nsig,np.dtype(dtype).name,"panel MB %.0f"%(x.nbytes/2**20),"steps_ms:"," ".join(res),flush=True)
        bx.free(); by.free(); dev.destroy()
