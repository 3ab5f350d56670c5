import sys; sys.path.insert(0,'.')  # run from the repo root
# does the placement of the panels in device memory change the step time?  (run-to-run spread of +-4 %
# between processes, none inside one process)
import numpy as np
import ctypes
from pygsp_amd import engine, graphs, filters, _capi
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
dtype=np.float64
x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
mode=sys.argv[1] if len(sys.argv)>1 else "grow"   # grow | none | big:<MB> | each:<MB>
pads=[]
for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 8):
    ctx=engine.Context(0)
    if mode=="grow" and trial: pads.append(ctx.alloc((trial*53+17)<<20))   # shifts every later allocation
    if mode.startswith("big:") and trial==0: pads.append(ctx.alloc(int(mode[4:])<<20))
    if mode.startswith("each:"): pads.append(ctx.alloc(int(mode[5:])<<20))
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
    dev.enable_gather_tiles()
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    ts=[]
    for _ in range(6):
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax); ts.append(ctx.last_timing()["steps_ms"]/30)
    wp=(ctypes.c_void_p*3)(); wb=(ctypes.c_int64*3)(); _capi.check(_capi.load().gspx_debug_workspace(ctx._h,wp,wb))
    print("ws_t %#x ws_r %#x vmm %d"%(wp[0] or 0,wp[1] or 0,wb[2]),end=" ")
    print(mode,"trial",trial,"x at %#x y at %#x"%(bx.ptr,by.ptr),"pads MB",sum(p.nbytes for p in pads)>>20,"ms/order %.4f"%min(ts),flush=True)
    bx.free(); by.free(); dev.destroy()
