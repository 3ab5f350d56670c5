import sys; sys.path.insert(0,'.')  # run from the repo root
# k_step_tile: odd steps walking each XCD's block range backwards (option alternate_sweep) - the tail of
# the previous step's panels is still in the 256 MB Infinity Cache when the next step starts there
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
    nodes,dn=filters.cheb_to_newton(c[0])
    dev.enable_gather_tiles()
    for nsig in (64,):
        x=np.random.default_rng(0).standard_normal((N,nsig)).astype(dtype)
        bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
        ys={}
        for rep in range(2):
          for alt in (2,6,10,14):
            ctx.set_option("alternate_sweep",1); ctx.set_option("tile_nt",alt-1)
            b1=b2=1e9
            for _ in range(4):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); b1=min(b1,ctx.last_timing()["steps_ms"]/30)
            ys[alt]=by.download((N,nsig),dtype)
            for _ in range(4):
                dev.newton_filter_dev(nodes,dn,bx.ptr,by.ptr,nsig,lmax); b2=min(b2,ctx.last_timing()["steps_ms"]/30)
            print(np.dtype(dtype).name,"nsig",nsig,"tile_nt",alt-1,"recurrence ms/order %.4f  newton %.4f"%(b1,b2),flush=True)
        print("   identical results:",bool(np.array_equal(ys[2],ys[14])))
        bx.free(); by.free()
    dev.destroy()
ctx.set_option("alternate_sweep",1); ctx.set_option("tile_nt",0)
