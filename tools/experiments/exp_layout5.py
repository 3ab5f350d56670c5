import sys, ctypes; sys.path.insert(0,'.')  # run from the repo root (GSPX_CONTIG=1: the reproducibly slow placement)
# which property of the block walk interacts with the placement of the panels?
import numpy as np
from pygsp_amd import engine, graphs, filters
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
dtype=np.float64
x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
ctx=engine.Context(0)
dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
lmax=2.0*float(dev.download_dw().max())
G=type("G",(),{"lmax":lmax,"e":None})()
c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
dev.enable_gather_tiles()
bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
y0=None
def t(**o):
    global y0
    for k,v in o.items(): ctx.set_option(k,v)
    b=1e9
    for _ in range(4):
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); b=min(b,ctx.last_timing()["steps_ms"]/30)
    y=by.download((N,64),dtype)
    if y0 is None: y0=y
    ok=bool(np.array_equal(y,y0))
    for k in o: ctx.set_option(k,0 if k!="alternate_sweep" else 1)
    return "%.4f%s"%(b,"" if ok else " (DIFFERENT RESULT)")
print("default",t(),flush=True)
for m in (1,2,4,3,5,7): print("tile_xcd_flip",m,t(tile_xcd_flip=m),flush=True)
for w in (496,480,448,384,256,640,768): print("tile_workgroups",w,t(tile_workgroups=w),flush=True)
print("alternate_sweep 0",t(alternate_sweep=0),flush=True)
for e in (1,2,4): print("tile_extra_every",e,t(tile_extra_every=e),flush=True)
print("tile_dynamic 1",t(tile_dynamic=1),flush=True)
print("default",t(),flush=True)
