// gspx_calib.hip.h - calibration kernel for graphs WITHOUT vertex locality (BASELINE configs 2 and 3):
// the rate at which this chip serves random row gathers.  Included by gspx.hip.
//
// On an Erdos-Renyi or block-model graph every stored entry of L fetches one panel row through an L2 miss
// (a 4 MB L2 against a 128-256 MB panel), so a recurrence step on such a graph is bound by
// `entries x max(row bytes, line)` over the rate of the L2-miss path, not by the algorithmic HBM bytes.  This
// kernel measures that rate directly, with nothing else in the way: rows of `row_bytes` (16 bytes per lane,
// row_bytes/16 lanes per row, 64/(row_bytes/16) rows per wave instruction - the lane layout of the step
// kernels) fetched from a panel by 32-bit indices, `in_flight` independent gathers per lane before the first
// use, no matrix values, no FMA, no panel writes.  The index stream is read coalesced (4 bytes per gather)
// and prefetched one block ahead.  Indices: uniform over the panel (ER), or block-local with probability
// p_intra (SBM: the panel cut into `blocks` equal row ranges, a gather issued "from" row i lands in i's own
// block with that probability, anywhere otherwise).  Workgroup w runs on XCD w % 8 and walks the w%8-th
// eighth of the gather stream, like the step kernels' XCD-contiguous row ranges, so block-local gathers of
// one XCD share that XCD's L2.
#pragma once

namespace gspx {

__device__ __forceinline__ unsigned long long calib_mix(unsigned long long z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_bench_gather_idx(unsigned* __restrict__ idx, long long n, unsigned panel_rows,
                                                          unsigned long long seed, int blocks, unsigned p_intra_u32) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long u = calib_mix(seed ^ ((unsigned long long)i * 0x100000001b3ull));
  unsigned r;
  if (blocks == 0) {
    // UNIQUE rows (counter calibration): an odd multiplier modulo a power of two is a bijection, so the first
    // panel_rows gathers touch every row exactly once, in a scattered order - known bytes, no reuse anywhere
    r = (unsigned)(((unsigned long long)i * 2654435761ull) & (unsigned long long)(panel_rows - 1));
  } else if (blocks > 1) {
    const unsigned bs = panel_rows / (unsigned)blocks;
    const unsigned from = (unsigned)(((unsigned __int128)(unsigned long long)i * panel_rows) / (unsigned long long)n);
    const unsigned blk = min(from / bs, (unsigned)blocks - 1);
    const unsigned coin = (unsigned)(calib_mix(u) >> 32);
    if (coin < p_intra_u32)
      r = blk * bs + (unsigned)(((u >> 32) * (unsigned long long)bs) >> 32);
    else
      r = (unsigned)(((u >> 32) * (unsigned long long)panel_rows) >> 32);
  } else {
    r = (unsigned)(((u >> 32) * (unsigned long long)panel_rows) >> 32);
  }
  idx[i] = r;
}

// LPR lanes x 16 bytes span a row; IF gathers in flight per lane
template <int LPR, int IF>
__global__ __launch_bounds__(256) void k_bench_gather(const uint4* __restrict__ panel, const unsigned* __restrict__ idx,
                                                      long long nblocks, unsigned* __restrict__ sink) {
  constexpr int R = 64 / LPR;  // rows per wave instruction
  constexpr long long PER = (long long)R * IF;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int xcd = blockIdx.x & 7;
  const long long wave_in_xcd = (long long)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long waves_per_xcd = (long long)(gridDim.x >> 3) * (blockDim.x >> 6);
  const long long lo = nblocks * xcd / 8, hi = nblocks * (xcd + 1) / 8;
  long long b = lo + wave_in_xcd;
  unsigned ix[IF];
  if (b < hi) {
#pragma unroll
    for (int j = 0; j < IF; ++j) ix[j] = idx[b * PER + j * R + g];
  }
  uint4 acc = {0u, 0u, 0u, 0u};
  while (b < hi) {
    uint4 v[IF];
#pragma unroll
    for (int j = 0; j < IF; ++j) v[j] = panel[(size_t)ix[j] * LPR + sub];
    const long long nb = b + waves_per_xcd;
    if (nb < hi) {  // the next block's indices travel while the rows are in flight
#pragma unroll
      for (int j = 0; j < IF; ++j) ix[j] = idx[nb * PER + j * R + g];
    }
#pragma unroll
    for (int j = 0; j < IF; ++j) {
      acc.x ^= v[j].x;
      acc.y += v[j].y;
      acc.z ^= v[j].z;
      acc.w += v[j].w;
    }
    b = nb;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = acc.x;  // keeps the loads alive
}

// Stream mix (round 6): NR read streams and NW write streams of `n4` 16-byte pieces each, walked together by a
// persistent grid - what the memory system delivers to a given read : write ratio with nothing else in the way.  The
// recurrence step is such a mix (T_{k-1} tiles, T_{k-2}, accumulator and entries in; T_k and the accumulator out: 3.6 : 1
// on a plain step, 2.3 : 1 on a flush step); these plain mixes run the same in the memory zones where the step runs
// 10 % apart (profiles/r06_placement.md) - they give the box's plain rates.  nt: bit 0 non-temporal loads, bit 1
// non-temporal stores, bit 2 write stream w goes IN PLACE over read stream w (the recurrence's T_{k-2} -> T_k), bit 3 the
// step's walk (XCD-partitioned 32 KB blocks) instead of a grid-stride sweep.
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_bench_streams(const u32x4* __restrict__ rd, u32x4* __restrict__ wr, size_t n4,
                                                       int nt, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  u32x4 acc = {0u, 0u, 0u, 0u};
  // bit 3: the WALK of the recurrence step instead of a grid-stride sweep - workgroup w belongs to "XCD" w % 8 and takes
  // 32 KB blocks of that XCD's eighth of every stream, strided over the XCD's workgroups: hundreds of separate 32 KB
  // regions of every stream are open at any time instead of one moving window
  const bool walk = (nt & 8) != 0;
  const size_t per_xcd = n4 / 8, nwx = gridDim.x >> 3, wx = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  constexpr size_t BLK = 2048;  // 16-byte pieces of a block
  const size_t total = walk ? (per_xcd / BLK + nwx - 1) / nwx * (BLK / 256) : (n4 + stride - 1) / stride;
  for (size_t it = 0; it < total; ++it) {
    size_t i;
    if (walk) {
      const size_t blk = wx + (it / (BLK / 256)) * nwx;
      i = xcd * per_xcd + blk * BLK + (it % (BLK / 256)) * 256 + threadIdx.x;
      if (blk * BLK >= per_xcd) break;
    } else {
      i = (size_t)blockIdx.x * 256 + threadIdx.x + it * stride;
      if (i >= n4) break;
    }
    u32x4 v[NR > 0 ? NR : 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const u32x4* src = rd + (size_t)r * n4 + i;
      v[r] = (nt & 1) ? __builtin_nontemporal_load(src) : *src;
    }
    u32x4 o = {(unsigned)i, 1u, 2u, 3u};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      o.x ^= v[r].x;
      o.y += v[r].y;
      o.z ^= v[r].z;
      o.w += v[r].w;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      u32x4* dst = ((nt & 4) && w < NR) ? (u32x4*)rd + (size_t)w * n4 + i : wr + (size_t)w * n4 + i;  // bit 2: in place
      if (nt & 2) __builtin_nontemporal_store(o, dst);
      else *dst = o;
    }
    if (NW == 0) {
      acc.x ^= o.x;
      acc.y += o.y;
    }
  }
  if (NW == 0 && (acc.x ^ acc.y) == 0x9e3779b9u) sink[0] = acc.x;  // keeps the loads alive
}

}  // namespace gspx

template <int NR>
static void launch_bench_streams(int nw, dim3 grid, hipStream_t st, const gspx::u32x4* rd, gspx::u32x4* wr, size_t n4, int nt,
                                 unsigned* sink) {
  switch (nw) {
    case 0: hipLaunchKernelGGL((gspx::k_bench_streams<NR, 0>), grid, dim3(256), 0, st, rd, wr, n4, nt, sink); break;
    case 1: hipLaunchKernelGGL((gspx::k_bench_streams<NR, 1>), grid, dim3(256), 0, st, rd, wr, n4, nt, sink); break;
    default: hipLaunchKernelGGL((gspx::k_bench_streams<NR, 2>), grid, dim3(256), 0, st, rd, wr, n4, nt, sink); break;
  }
}

extern "C" int gspx_bench_streams(gspx_ctx* ctx, int64_t bytes_per_stream, int n_read, int n_write, int nt,
                                  int workgroups_per_cu, int iters, double* gbps) {
  if (!ctx || !gbps || bytes_per_stream < 4096 || n_read < 0 || n_read > 4 || n_write < 0 || n_write > 2 ||
      n_read + n_write < 1 || iters < 1 || workgroups_per_cu < 1 || workgroups_per_cu > 16 || nt < 0 || nt > 31)
    return set_err(GSPX_ERR_INVALID, "gspx_bench_streams: bad argument (0-4 read streams, 0-2 write streams)");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n4 = (size_t)bytes_per_stream / 16;
  DevMem rd, wr, sink;
  const bool in_ws = (nt & 16) != 0;  // bit 4: the streams lie in the context's T workspace (whatever backs it) instead of
  if (in_ws) {                        // fresh allocations - the memory the recurrence itself streams
    if (!ctx->ws_t.p || ctx->ws_t.bytes < n4 * 16 * (size_t)(n_read + n_write))
      return set_err(GSPX_ERR_INVALID, "gspx_bench_streams: the context's workspace is smaller than the streams");
  } else {
    CHK(rd.alloc(std::max<size_t>(n4 * 16 * (size_t)n_read, 64)));
    CHK(wr.alloc(std::max<size_t>(n4 * 16 * (size_t)n_write, 64)));
  }
  CHK(sink.alloc(64));
  const gspx::u32x4* rdp = in_ws ? (const gspx::u32x4*)ctx->ws_t.p : (const gspx::u32x4*)rd.p;
  gspx::u32x4* wrp = in_ws ? (gspx::u32x4*)ctx->ws_t.p + n4 * (size_t)n_read : (gspx::u32x4*)wr.p;
  if (n_read)
    hipLaunchKernelGGL((k_fill<float>), dim3(4096), dim3(256), 0, st, (float*)rdp, n4 * 4 * (size_t)n_read, 1.0f);
  const dim3 grid((unsigned)(ctx->cu_count * workgroups_per_cu));
  auto launch = [&]() {
    switch (n_read) {
      case 0: launch_bench_streams<0>(n_write, grid, st, rdp, wrp, n4, nt & 15, sink.as<unsigned>()); break;
      case 1: launch_bench_streams<1>(n_write, grid, st, rdp, wrp, n4, nt & 15, sink.as<unsigned>()); break;
      case 2: launch_bench_streams<2>(n_write, grid, st, rdp, wrp, n4, nt & 15, sink.as<unsigned>()); break;
      case 3: launch_bench_streams<3>(n_write, grid, st, rdp, wrp, n4, nt & 15, sink.as<unsigned>()); break;
      default: launch_bench_streams<4>(n_write, grid, st, rdp, wrp, n4, nt & 15, sink.as<unsigned>()); break;
    }
  };
  launch();  // warm-up
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  for (int i = 0; i < iters; ++i) launch();
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  float t = 0;
  HIPCHK(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
  *gbps = (double)n4 * 16.0 * (n_read + n_write) * iters / ((double)t * 1e-3) / 1e9;
  return GSPX_OK;
}

template <int LPR>
static void launch_bench_gather(int in_flight, dim3 grid, hipStream_t st, const uint4* panel, const unsigned* idx,
                                long long nblocks, unsigned* sink) {
  switch (in_flight) {
    case 2: hipLaunchKernelGGL((gspx::k_bench_gather<LPR, 2>), grid, dim3(256), 0, st, panel, idx, nblocks, sink); break;
    case 4: hipLaunchKernelGGL((gspx::k_bench_gather<LPR, 4>), grid, dim3(256), 0, st, panel, idx, nblocks, sink); break;
    case 8: hipLaunchKernelGGL((gspx::k_bench_gather<LPR, 8>), grid, dim3(256), 0, st, panel, idx, nblocks, sink); break;
    default: hipLaunchKernelGGL((gspx::k_bench_gather<LPR, 16>), grid, dim3(256), 0, st, panel, idx, nblocks, sink); break;
  }
}

extern "C" int gspx_bench_gather(gspx_ctx* ctx, int64_t panel_rows, int row_bytes, int64_t n_gathers, int in_flight,
                                 int blocks, double p_intra, int workgroups_per_cu, int iters, double* ms, double* gbps) {
  if (!ctx || panel_rows < 1 || panel_rows >= ((int64_t)1 << 32) || n_gathers < 1 || iters < 1 || blocks < 0 ||
      blocks > panel_rows || !(p_intra >= 0.0 && p_intra <= 1.0))
    return set_err(GSPX_ERR_INVALID, "gspx_bench_gather: bad argument");
  if (blocks == 0 && ((panel_rows & (panel_rows - 1)) != 0 || n_gathers > panel_rows))
    return set_err(GSPX_ERR_INVALID, "gspx_bench_gather: unique rows (blocks == 0) need a power-of-two panel_rows >= n_gathers");
  if (!(row_bytes == 64 || row_bytes == 128 || row_bytes == 256 || row_bytes == 512))
    return set_err(GSPX_ERR_INVALID, "gspx_bench_gather: row_bytes must be 64, 128, 256 or 512");
  if (!(in_flight == 2 || in_flight == 4 || in_flight == 8 || in_flight == 16))
    return set_err(GSPX_ERR_INVALID, "gspx_bench_gather: in_flight must be 2, 4, 8 or 16");
  if (workgroups_per_cu < 1 || workgroups_per_cu > 16)
    return set_err(GSPX_ERR_INVALID, "gspx_bench_gather: workgroups_per_cu must be in [1, 16]");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int lpr = row_bytes / 16;
  const int64_t per = (int64_t)(64 / lpr) * in_flight;
  const int64_t nblocks = (n_gathers + per - 1) / per;  // rounded up to whole wave iterations
  const int64_t n = nblocks * per;
  DevMem panel, idx, sink;
  CHK(panel.alloc((size_t)panel_rows * row_bytes));
  CHK(idx.alloc((size_t)n * sizeof(unsigned)));
  CHK(sink.alloc(64));
  hipLaunchKernelGGL((k_fill<float>), dim3(4096), dim3(256), 0, st, panel.as<float>(),
                     (size_t)panel_rows * row_bytes / 4, 1.0f);
  const double pu = p_intra * 4294967296.0;
  hipLaunchKernelGGL(gspx::k_bench_gather_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx.as<unsigned>(),
                     (long long)n, (unsigned)panel_rows, 0x5eedull, blocks,
                     (unsigned)std::min(pu, 4294967295.0));
  const dim3 grid((unsigned)(ctx->cu_count * workgroups_per_cu / 8 * 8));
  auto launch = [&]() {
    switch (lpr) {
      case 4: launch_bench_gather<4>(in_flight, grid, st, (const uint4*)panel.p, idx.as<unsigned>(), nblocks, sink.as<unsigned>()); break;
      case 8: launch_bench_gather<8>(in_flight, grid, st, (const uint4*)panel.p, idx.as<unsigned>(), nblocks, sink.as<unsigned>()); break;
      case 16: launch_bench_gather<16>(in_flight, grid, st, (const uint4*)panel.p, idx.as<unsigned>(), nblocks, sink.as<unsigned>()); break;
      default: launch_bench_gather<32>(in_flight, grid, st, (const uint4*)panel.p, idx.as<unsigned>(), nblocks, sink.as<unsigned>()); break;
    }
  };
  launch();  // warm-up
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  for (int i = 0; i < iters; ++i) launch();
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  float t = 0;
  HIPCHK(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
  const double per_launch = (double)t / iters;
  if (ms) *ms = per_launch;
  if (gbps) *gbps = (double)n * row_bytes / (per_launch * 1e-3) / 1e9;
  return GSPX_OK;
}
