import sys; sys.path.insert(0,'.')  # run from the repo root
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
G=graphs.Sensor(100000,seed=42); G.estimate_lmax("bounds"); dev=G.device_graph(); lmax=G.lmax
c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=30))
for nsig in (1,2,4):
    x=np.random.default_rng(0).standard_normal((G.N,nsig)); bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    for rpw in (1,2,4):
        for g in (0,1,2,3):
            ctx.set_option("rows_per_wave",rpw); ctx.set_option("narrow_g_log2",g)
            best=1e9
            for _ in range(10):
                dev.cheby_filter_dev(c,bx.ptr,by.ptr,nsig,lmax); t=ctx.last_timing(); best=min(best,t["total_ms"])
            print("nsig",nsig,"rpw",rpw,"glog2",g,"total ms %.4f"%best,"per step us %.2f"%(t["steps_ms"]/30*1e3),flush=True)
