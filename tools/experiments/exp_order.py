import sys; sys.path.insert(0,'.')  # run from the repo root
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
orders={"morton":engine.locality_order(W,coords),"hilbert":engine.hilbert_order(coords),"rcm":engine.locality_order(W,None)}
for dtype in (np.float64,np.float32):
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    for name,perm in orders.items():
        dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
        lmax=2.0*float(dev.download_dw().max())
        G=type("G",(),{"lmax":lmax,"e":None})()
        c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
        nodes,d=filters.cheb_to_newton(c[0])
        b1=b2=1e9
        for _ in range(3):
            dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); b1=min(b1,t["steps_ms"]/t["step_launches"])
            dev.newton_filter_dev(nodes,d,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); b2=min(b2,t["steps_ms"]/t["step_launches"])
        print(np.dtype(dtype).name,name,"recurrence ms %.4f"%b1,"newton ms %.4f"%b2,flush=True)
        dev.destroy()
    bx.free(); by.free()
