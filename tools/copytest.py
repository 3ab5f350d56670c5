import sys; sys.path.insert(0,'.')
from pygsp_amd import engine
ctx = engine.default_context(0)
for mb in (4, 8, 16, 32, 64, 128, 512, 2048):
    print(mb, "MB x2 buffers:", round(ctx.bench_copy(mb << 20, max(5, 4000 // mb)), 1), "GB/s")
