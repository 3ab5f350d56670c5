import sys, time; sys.path.insert(0,'.')  # run from the repo root
# LDS-staged recurrence step (k_step_tile) vs the plain gather kernels on the headline workload
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
WGS=[int(a) for a in sys.argv[1:]] or [0]
for dtype in (np.float32,np.float64):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by,bz=ctx.upload(x),ctx.alloc(x.nbytes),ctx.alloc(x.nbytes)
    t0=time.time(); st=dev.enable_gather_tiles(); print("tiles",st,"build s %.1f"%(time.time()-t0),flush=True)
    for tg,wg in [(0,0)]+[(1,w) for w in WGS]:
        ctx.set_option("tile_gather",tg); ctx.set_option("tile_workgroups",wg)
        best=1e9
        for _ in range(3):
            dev.cheby_filter_dev(c,bx.ptr,(by if tg else bz).ptr,64,lmax); t=ctx.last_timing(); best=min(best,t["steps_ms"]/30)
        print(np.dtype(dtype).name,"tile_gather",tg,"workgroups",wg,"ms per order %.4f"%best,flush=True)
    nodes,d=filters.cheb_to_newton(c[0])
    for tg in (0,1):
        ctx.set_option("tile_gather",tg); ctx.set_option("tile_workgroups",0)
        best=1e9
        for _ in range(3):
            dev.newton_filter_dev(nodes,d,bx.ptr,by.ptr,64,lmax); t=ctx.last_timing(); best=min(best,t["steps_ms"]/30)
        print(np.dtype(dtype).name,"NEWTON tile_gather",tg,"ms per order %.4f"%best,flush=True)
    ctx.set_option("tile_gather",1)
    dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)
    y1=by.download((N,64),dtype); y0=bz.download((N,64),dtype)
    print("max rel diff tile vs plain %.2e"%(np.max(abs(y1-y0))/np.max(abs(y0))),flush=True)
    ctx.set_option("tile_gather",1); ctx.set_option("tile_workgroups",0)
    bx.free(); by.free(); bz.free(); dev.destroy()
