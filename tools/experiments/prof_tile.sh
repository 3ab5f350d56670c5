# PMC counters of the fused Newton-pair kernel vs the single-step Newton kernel (f32 headline graph).
# usage (GPU box): bash tools/experiments/prof_tile.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_tile
mkdir -p $OUT
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cat > /tmp/pair_once.py <<PY
import sys; sys.path.insert(0,'$R')
import numpy as np
from pygsp_amd import engine, graphs, filters
ctx=engine.default_context(0)
N=1000000
W,coords=graphs.sensor_weights(N,k=8,seed=42)
perm=engine.locality_order(W,coords)
for dtype in (np.float32,np.float64):
    dev=engine.DeviceGraph.from_w(W,dtype=dtype,perm=perm,ctx=ctx)
    lmax=2.0*float(dev.download_dw().max())
    G=type("G",(),{"lmax":lmax,"e":None})()
    c=np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G,50),m=6))
    x=np.random.default_rng(0).standard_normal((N,64)).astype(dtype)
    bx,by=ctx.upload(x),ctx.alloc(x.nbytes)
    dev.enable_gather_tiles()
    for tg in (1,0):
        ctx.set_option("tile_gather",tg)
        dev.cheby_filter_dev(c,bx.ptr,by.ptr,64,lmax)
    bx.free(); by.free(); dev.destroy()
PY
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64 SQ_INST_CYCLES_VMEM" ; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python /tmp/pair_once.py > /dev/null 2> $OUT/pmc_$name.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python /tmp/pair_once.py > /dev/null 2> $OUT/stats.err
python - <<PY
import csv, glob, os
from collections import defaultdict
out="$OUT"
for f in sorted(glob.glob(os.path.join(out,"stats/**/*kernel_stats.csv"),recursive=True)):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("{:60.60s} calls={:>4} avg_ns={:>10}".format(r["Name"],r["Calls"],r["AverageNs"]))
for f in sorted(glob.glob(os.path.join(out,"pmc_*/**/*counter_collection.csv"),recursive=True)):
    acc=defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"] or 0))
    for k,cs in acc.items():
        if "k_step" not in k: continue
        for cn,v in cs.items():
            print("{:50.50s} {:28s} n={:3d} mean={:.6g}".format(k,cn,len(v),sum(v)/len(v)))
PY
