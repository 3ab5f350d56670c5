import sys, json; sys.path.insert(0,'.')  # run from the repo root
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float64,np.float32):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    ctx.set_option("kernel",1)
    for wpb in (4,8,16):
        for rpw in (2,4,8):
            ctx.set_option("waves_per_block",wpb); ctx.set_option("rows_per_wave",rpw)
            best=1e9
            for _ in range(3):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); best=min(best,t["steps_ms"]/t["step_launches"])
            print(np.dtype(dtype).name,"wpb",wpb,"rpw",rpw,"ms %.4f"%best,flush=True)
    ctx.set_option("kernel",0); ctx.set_option("waves_per_block",4); ctx.set_option("rows_per_wave",0)
    bx.free(); by.free(); dev.destroy()
