import sys, ctypes; sys.path.insert(0,'.')  # run from the repo root
# same allocation, different distance between the two T_k slots (grow-only workspace: allocate with the
# largest gap first, then shrink the gap without reallocating)
import numpy as np
from pygsp_amd import engine, graphs, filters, _capi
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
dtype=np.float64
x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
MB=1<<20
keep=[]
for attempt in range(5):
    ctx=engine.Context(0)
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    dev.enable_gather_tiles()
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    def t():
        b=1e9
        for _ in range(4):
            dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); b=min(b,ctx.last_timing()["steps_ms"]/30)
        return b
    def ws():
        wp=(ctypes.c_void_p*3)(); wb=(ctypes.c_int64*3)(); _capi.check(_capi.load().gspx_debug_workspace(ctx._h,wp,wb)); return wp[0] or 0
    res=[]
    for gap in (160*MB, 128*MB, 64*MB+4096, 24*MB, 16*MB, 2*MB, 256*1000, 0):
        ctx.set_option("panel_gap",gap)
        res.append("%d:%.4f"%(gap>>10,t()))
    print("context",attempt,"ws_t %#x"%ws(),"gapKiB:ms"," ".join(res),flush=True)
    keep.append((ctx,dev,bx,by))
