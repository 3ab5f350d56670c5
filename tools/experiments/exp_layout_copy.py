import sys; sys.path.insert(0,'.')  # run from the repo root; GSPX_CONTIG / GSPX_VMM_CHUNK_MB select the allocator
# does a plain streaming copy see the placement too?
from pygsp_amd import engine
for trial in range(4):
    ctx=engine.Context(0)
    r=[ctx.bench_copy(1<<30,20) for _ in range(2)]+[ctx.bench_read(2<<30,4)]
    print("context",trial,"copy 2x1GiB GB/s %.0f %.0f  read 2GiB GB/s %.0f"%tuple(r),flush=True)
