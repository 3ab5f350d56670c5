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
    nodes,d=filters.cheb_to_newton(c[0])
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    for alt in (0,1):
        for remap in (1,0):
            ctx.set_option("alternate_sweep",alt); ctx.set_option("xcd_remap",remap)
            b1=b2=1e9
            for _ in range(3):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); b1=min(b1,t["steps_ms"]/t["step_launches"])
                dev.newton_filter_dev(nodes,d,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); b2=min(b2,t["steps_ms"]/t["step_launches"])
            print(np.dtype(dtype).name,"alt",alt,"remap",remap,"recurrence ms %.4f"%b1,"newton ms %.4f"%b2,flush=True)
    ctx.set_option("alternate_sweep",0); ctx.set_option("xcd_remap",1)
    bx.free(); by.free(); dev.destroy()
