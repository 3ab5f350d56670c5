import sys, time, ctypes; sys.path.insert(0,'.')  # run from the repo root
# per-workgroup entry/exit clocks of k_step_tile on the headline workload: how much of a launch is
# ramp-up, tail and imbalance between the persistent workgroups (static block partition)?
import numpy as np
from pygsp_amd import engine, graphs, filters, _capi
ctx=engine.default_context(0)
lib=_capi.load()
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
print("tiles",dev.enable_gather_tiles(),flush=True)
nodes,dn=filters.cheb_to_newton(c[0])
def run(newton):
    if newton: dev.newton_filter_dev(nodes,dn,bx.ptr,by.ptr,64,lmax)
    else: dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)
y_ref=None
MODES=[("static",dict(tile_dynamic=0))]+[("extra every %d"%e,dict(tile_dynamic=0,tile_extra_every=e)) for e in (4,6,8,10,14)]+[
       ("prio younger",dict(tile_dynamic=0,tile_prio=1)),("prio alternate",dict(tile_dynamic=0,tile_prio=2)),("tickets",dict(tile_dynamic=1)),("static",dict(tile_dynamic=0))]
for name,opts in MODES:
  for newton in (0,1):
    for kk in ("tile_dynamic","tile_extra_every","tile_prio","tile_stamps"): ctx.set_option(kk,0)
    for kk,v in opts.items(): ctx.set_option(kk,v)
    best=1e9
    for _ in range(4):
        run(newton); best=min(best,ctx.last_timing()["steps_ms"]/30)
    y=by.download((N,64),dtype)
    if newton==0:
        if y_ref is None: y_ref=y
        assert np.array_equal(y,y_ref),name
    ctx.set_option("tile_stamps",1)
    nl=ctypes.c_int64(); nw=ctypes.c_int()
    run(newton)
    _capi.check(lib.gspx_debug_tile_stamps(ctx._h,None,0,ctypes.byref(nl),ctypes.byref(nw)))
    run(newton)
    buf=np.zeros((30,nw.value,2),dtype=np.int64)
    _capi.check(lib.gspx_debug_tile_stamps(ctx._h,_capi.ptr(buf),buf.size,ctypes.byref(nl),ctypes.byref(nw)))
    d=(buf[:,:,1]-buf[:,:,0])*0.01; sp=(buf[:,:,1].max(1)-buf[:,:,0].min(1))*0.01
    busy=(d.sum(1)/(nw.value*sp)).mean()
    o=np.argsort(sp); pl=o[:15]; fl=o[-8:]
    h=nw.value//2
    print("%-16s newton %d ms/order %.4f busy %.3f | plain span %.1f first-half WGs %.1f second-half %.1f | flush span %.1f first %.1f second %.1f"%(
        name,newton,best,busy,np.median(sp[pl]),d[pl][:,:h].mean(),d[pl][:,h:].mean(),np.median(sp[fl]),d[fl][:,:h].mean(),d[fl][:,h:].mean()),flush=True)
