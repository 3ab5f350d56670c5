import sys; sys.path.insert(0,'.')  # run from the repo root
# hipGraph replay of a repeated call on launch-bound (cache-resident) graphs: BASELINE config 1 and friends
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
for N,nsig in ((100000,1),(100000,8),(30000,1),(1000000,1)):
    W,coords=graphs.sensor_weights(N,k=6,seed=42)
    dev=engine.DeviceGraph.from_w(W,perm=engine.locality_order(W,coords),ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    x=np.random.default_rng(0).standard_normal((N,nsig))
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    for gl in (0,1):
        ctx.set_option("graph_launch",gl)
        best=1e9
        for _ in range(20):
            dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); best=min(best,ctx.last_timing()["total_ms"])
        print("N",N,"nsig",nsig,"graph_launch",gl,"total ms %.4f"%best,"per launch us %.2f"%(best/31*1e3),flush=True)
    ctx.set_option("graph_launch",2)
    bx.free(); by.free(); dev.destroy()
