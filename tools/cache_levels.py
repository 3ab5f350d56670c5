"""Calibration: read-only GB/s by footprint (L2 / Infinity Cache / HBM) and streaming-copy GB/s of this
box, with the engine's own kernels (gspx_bench_read / gspx_bench_copy).  GPU box only."""
import sys; sys.path.insert(0,'.')
from pygsp_amd import engine
ctx = engine.default_context(0)
print("read-only, whole chip streaming the same buffer (GB/s):")
for kb in (16, 256, 1024, 2048, 4096, 8192, 16384, 65536, 131072, 262144, 1048576, 4194304):
    passes = max(2, min(400, (8 << 30) // (kb << 10)))
    print("  %8d KiB x %4d passes: %8.0f" % (kb, passes, ctx.bench_read(kb << 10, passes)))
print("copy (read+write) GB/s:")
for mb in (16, 64, 128, 512, 2048):
    print("  %5d MiB x2 buffers: %8.0f" % (mb, ctx.bench_copy(mb << 20, max(5, 4000 // mb))))
