import sys, ctypes; sys.path.insert(0,'.')  # run from the repo root
# are the per-workgroup durations of k_step_tile (static walk) the same from launch to launch?
import numpy as np
from pygsp_amd import engine, graphs, filters, _capi
ctx=engine.default_context(0); lib=_capi.load()
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
dtype=np.float64 if "f32" not in sys.argv else np.float32
dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
lmax=2.0*float(dev.download_dw().max())
G=type("G",(),{"lmax":lmax,"e":None})()
c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
dev.enable_gather_tiles()
ctx.set_option("tile_dynamic",0)
for _ in range(2): dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)
ctx.set_option("tile_stamps",1)
nl=ctypes.c_int64(); nw=ctypes.c_int()
durs=[]
for rep in range(2):
    dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)
    _capi.check(lib.gspx_debug_tile_stamps(ctx._h,None,0,ctypes.byref(nl),ctypes.byref(nw))) if rep==0 and False else None
    buf=np.zeros((30,512,2),dtype=np.int64)
    _capi.check(lib.gspx_debug_tile_stamps(ctx._h,_capi.ptr(buf),buf.size,ctypes.byref(nl),ctypes.byref(nw)))
    durs.append((buf[:,:,1]-buf[:,:,0])*0.01)
d=np.concatenate(durs)  # [60][512]
plain=[l for l in range(60) if d[l].mean()<1.25*np.median(d.mean(axis=1)) and (l%30)>1]
flush=[l for l in range(60) if d[l].mean()>1.25*np.median(d.mean(axis=1))]
P=d[plain]; F=d[flush]
def corr(A):
    C=np.corrcoef(A); return C[np.triu_indices(len(A),1)].mean()
print("plain launches",len(plain),"mean pairwise corr of per-WG durations %.3f"%corr(P))
print("flush launches",len(flush),"mean pairwise corr %.3f"%corr(F))
print("corr(mean plain profile, mean flush profile) %.3f"%np.corrcoef(P.mean(0),F.mean(0))[0,1])
m=P.mean(0); print("plain per-WG mean: min %.1f mean %.1f max %.1f; std of profile %.2f, residual std %.2f"%(m.min(),m.mean(),m.max(),m.std(),(P-m).std()))
# structure: by XCD, by slot within XCD (blockIdx>>3), by the 30/31-block split
wg=np.arange(512); xcd=wg&7; slot=wg>>3
print("by xcd:"," ".join("%.1f"%m[xcd==i].mean() for i in range(8)))
per_xcd=(15625+7)//8
nblk=np.array([len(range(s,min(per_xcd,15625-x*per_xcd),64)) for x,s in zip(xcd,slot)])
for n in np.unique(nblk): print("WGs with",n,"blocks:",(nblk==n).sum(),"mean dur %.1f"%m[nblk==n].mean())
print("by slot (8 groups of 8):"," ".join("%.1f"%m[(slot>>3)==i].mean() for i in range(8)))
