import sys; sys.path.insert(0,'.')  # run from the repo root
# find a context in the slow placement state, then move the panels around inside their workspaces
import numpy as np
from pygsp_amd import engine, graphs, filters
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
dtype=np.float64
x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
def setup():
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
    return ctx,dev,bx,by,t
keep=[]
for attempt in range(6):
    ctx,dev,bx,by,t=setup()
    base=t()
    print("context",attempt,"baseline %.4f"%base,flush=True)
    if base>0.38: break
    keep.append((ctx,dev,bx,by))   # keep its memory: the next context lands elsewhere
else:
    print("no slow state found"); sys.exit(0)
MB=1<<20
for gap in (256,4096,65536,MB,2*MB,16*MB+4096,64*MB,100*MB+256*7):
    ctx.set_option("panel_gap",gap); ctx.set_option("racc_shift",0)
    print("panel_gap %10d: %.4f"%(gap,t()),flush=True)
ctx.set_option("panel_gap",0)
for sh in (256,4096,65536,MB,2*MB,16*MB+4096,64*MB,100*MB+256*7):
    ctx.set_option("racc_shift",sh)
    print("racc_shift %10d: %.4f"%(sh,t()),flush=True)
ctx.set_option("racc_shift",0)
# new x / y buffers at other addresses
for i in range(3):
    pad=ctx.alloc((37+i*29)*MB)
    bx2,by2=ctx.upload(x),ctx.alloc(x.nbytes)
    b=1e9
    for _ in range(4):
        dev.cheby_filter_dev(dev_c if False else np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(type("G",(),{"lmax":2.0*float(dev.download_dw().max()),"e":None})(),50),m=30)),bx2.ptr,by2.ptr,64,2.0*float(dev.download_dw().max())); b=min(b,ctx.last_timing()["steps_ms"]/30)
    print("other x/y buffers (%#x %#x): %.4f"%(bx2.ptr,by2.ptr,b),flush=True)
