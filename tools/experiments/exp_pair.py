import sys, time; sys.path.insert(0,'.')  # run from the repo root
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
PAIRS=(0,1)
REMAP=[int(a) for a in sys.argv[1:]] or [1]
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float32,np.float64):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    nodes,d=filters.cheb_to_newton(c[0])
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by,bz=ctx.upload(x),ctx.alloc(x.nbytes),ctx.alloc(x.nbytes)
    t0=time.time(); st=dev.enable_pair_tiles(); print("tiles",st,"build s %.1f"%(time.time()-t0),flush=True)
    for pair,remap in [(0,1)]+[(1,r) for r in REMAP]:
        ctx.set_option("newton_pair",pair); ctx.set_option("xcd_remap",remap)
        best=1e9
        for _ in range(3):
            dev.newton_filter_dev(nodes,d,bx.ptr,(by if pair else bz).ptr,64,lmax); t=ctx.last_timing(); best=min(best,t["steps_ms"]/30)
        print(np.dtype(dtype).name,"pair",pair,"xcd_remap",remap,"ms per order %.4f"%best,flush=True)
    y1=by.download((N,64),dtype); y0=bz.download((N,64),dtype)
    print("max rel diff pair vs single %.2e"%(np.max(abs(y1-y0))/np.max(abs(y0))),flush=True)
    bx.free(); by.free(); bz.free(); dev.destroy()
