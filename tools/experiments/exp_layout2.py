import sys; sys.path.insert(0,'.')  # run from the repo root
# placement of the panels: gaps between the T_k slots / offset of the accumulator, one process
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
nodes,dn=filters.cheb_to_newton(c[0])
dev.enable_gather_tiles()
bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
def t(newton=False):
    b=1e9
    for _ in range(4):
        (dev.newton_filter_dev(nodes,dn,bx.ptr,by.ptr,64,lmax) if newton else dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)); b=min(b,ctx.last_timing()["steps_ms"]/30)
    return b
GAPS=[0,256,1024,4096,16384,65536,1<<20,(2<<20)+4096,12345*256]
for gap in GAPS:
    ctx.set_option("panel_gap",gap); ctx.set_option("racc_shift",0)
    print("panel_gap %9d racc_shift 0: %.4f"%(gap,t()),flush=True)
ctx.set_option("panel_gap",0)
for sh in GAPS[1:]:
    ctx.set_option("racc_shift",sh)
    print("panel_gap 0 racc_shift %9d: %.4f"%(sh,t()),flush=True)
