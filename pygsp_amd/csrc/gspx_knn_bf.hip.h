// gspx_knn_bf.hip.h - k nearest neighbours in MORE than three dimensions (NNGraph on feature / patch clouds,
// pygsp/graphs/nngraphs/nngraph.py:213-226, nngraphs/imgpatches.py): tiled brute force with the pair
// distances on the matrix cores.  Included by gspx_knn.hip.h (uses its KD-tree arithmetic helpers).
//
// A uniform grid stops paying beyond three dimensions (a KD-tree degrades the same way), while all pair
// distances of a point block are ONE dense contraction,  |x - y|^2 = |x|^2 + |y|^2 - 2 x.y  - the one place
// of the path's surroundings where a dense panel product is the natural formulation, so it runs on MFMA
// (v_mfma_f64_16x16x4f64: a 16 x 16 tile of dot products per instruction, fp64 because the selection below
// must not lose a true neighbour to rounding).  Four stages:
//   1. k_bf_prepare   points -> MFMA operand order (per 16-point tile and 4 dimensions: one coalesced 512-byte
//                     line, lane l = point l % 16, dimension l / 16), squared norms
//   2. k_bf_tau       per query an UPPER BOUND tau of its k-th neighbour distance: the exact k-th smallest
//                     squared distance to a strided sample of M points (a subset's order statistics bound the
//                     full set's from above)
//   3. k_bf_collect   MFMA sweep of every 64-query block over the points: every pair whose approximate squared
//                     distance is <= tau (+ a rounding margin) is appended to the query's candidate list.  Two
//                     sweeps: a strided subset of sqrt(N M) points, then - after k_bf_refine tightened every bound to the
//                     exact k-th smallest of the candidates so far - the rest: k (N1 / M + N / N1) entries per query
//   4. k_bf_select    candidates re-evaluated in the KD-tree's own arithmetic (knn_sqdist: per-dimension
//                     differences, no fused multiply-add; correctly rounded square root) and the k smallest by
//                     (distance, index) kept - so neighbours and distances equal scipy's bit for bit, exactly
//                     like the grid search in 1-3 dimensions.  A query whose list overflowed (heavy ties /
//                     duplicates) scans all points exactly instead.
// Other metrics (manhattan, max_dist) have no product form: every query takes the exact scan of stage 4.
#pragma once

namespace gspx {

typedef double bf_d4 __attribute__((ext_vector_type(4)));

// knn_sqdist / knn_key (the KD-tree's arithmetic, gspx_knn.hip.h) with the query in REGISTERS: q is indexed with
// compile-time constants only (4 DT >= d values, fully unrolled, the real dimension d guards each block), so it
// never goes to scratch memory - the runtime-indexed q[64] of the first version did, and a scratch read per
// multiply made these kernels several times slower than their arithmetic.
template <int DT>
__device__ __forceinline__ double knn_sqdist_reg(const double (&q)[4 * DT], const double* __restrict__ p, int d) {
#pragma clang fp contract(off)
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
  for (int b = 0; b < DT; ++b)
    if (4 * b + 4 <= d) {
      const double d0 = q[4 * b] - p[4 * b], d1 = q[4 * b + 1] - p[4 * b + 1], d2 = q[4 * b + 2] - p[4 * b + 2],
                   d3 = q[4 * b + 3] - p[4 * b + 3];
      const double s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
      a0 = a0 + s0;
      a1 = a1 + s1;
      a2 = a2 + s2;
      a3 = a3 + s3;
    }
  double s = ((a0 + a1) + a2) + a3;
#pragma unroll
  for (int b = 0; b < DT; ++b)
    if (4 * b < d && 4 * b + 4 > d) {  // the block that holds the last d % 4 dimensions
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (4 * b + r < d) {
          const double df = q[4 * b + r] - p[4 * b + r];
          const double sq = df * df;
          s = s + sq;
        }
    }
  return s;
}
template <int DT>
__device__ __forceinline__ double knn_key_reg(const double (&q)[4 * DT], const double* __restrict__ p, int d, int metric) {
  if (metric == 0) return knn_sqdist_reg<DT>(q, p, d);
  double s = 0;
#pragma unroll
  for (int j = 0; j < 4 * DT; ++j)
    if (j < d) {
      const double a = fabs(q[j] - p[j]);
      s = metric == 1 ? s + a : fmax(s, a);
    }
  return s;
}
template <int DT> __device__ __forceinline__ void bf_load_query(double (&q)[4 * DT], const double* __restrict__ row, int d) {
#pragma unroll
  for (int j = 0; j < 4 * DT; ++j) q[j] = j < d ? row[j] : 0.0;
}

// X (N x d, row major) -> Xop[tile][t][lane] = X[16 tile + (lane & 15)][4 t + (lane >> 4)] (zero beyond N / d),
// norm[i] = sum_j x_ij^2 (padded to whole tiles with 1e300)
__global__ void k_bf_prepare(const double* __restrict__ x, int N, int d, int DT, double* __restrict__ xop,
                             double* __restrict__ norm, float* __restrict__ xop32) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)((N + 15) / 16) * DT * 64;
  if (gid < total) {
    const int lane = (int)(gid & 63);
    const long long tt = gid >> 6;
    const int t = (int)(tt % DT);
    const long long tile = tt / DT;
    const long long p = tile * 16 + (lane & 15);
    const int j = 4 * t + (lane >> 4);
    const double v = (p < N && j < d) ? x[p * d + j] : 0.0;
    xop[gid] = v;
    if (xop32) xop32[gid] = (float)v;  // operands of the fp32 sweep (the bound's margin covers their rounding)
  }
  if (gid < N) {
    double s = 0;
    for (int j = 0; j < d; ++j) s += x[gid * d + j] * x[gid * d + j];
    norm[gid] = s;
  } else if (gid < (long long)((N + 15) / 16) * 16) {
    norm[gid] = 1e300;  // the padding points of the last tile are infinitely far away
  }
}

// tau[i] = k-th smallest exact key (squared euclidean distance in the KD-tree's arithmetic) from point i to the
// sample {0, stride, 2 stride, ...} without i itself: an upper bound of its k-th neighbour's key
template <int KMAX, int DT>
__global__ __launch_bounds__(128) void k_bf_tau(const double* __restrict__ x, int N, int d, int k, int stride,
                                                double* __restrict__ tau) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  double q[4 * DT];
  bf_load_query<DT>(q, x + (size_t)i * d, d);
  double bd[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) bd[t] = 1e300;
  for (int c = 0; c < N; c += stride) {
    if (c == i) continue;
    double cd = knn_sqdist_reg<DT>(q, x + (size_t)c * d, d);
    if (cd < bd[KMAX - 1]) {
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const bool lt = cd < bd[t];
        const double td = bd[t];
        bd[t] = lt ? cd : td;
        cd = lt ? td : cd;
      }
    }
  }
  double kth = 1e300;
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t == k - 1) kth = bd[t];
  tau[i] = kth;
}

// candidate lists: either N fixed-capacity rows (off == null; cap == 0: count only), or rows of their own lengths
// (off[q] .. off[q + 1], from a counting sweep: the radius search, whose neighbourhoods have no common bound)
__device__ __forceinline__ void bf_append(const int* __restrict__ off, int cap, int q, int slot, int c,
                                          int* __restrict__ buf) {
  if (off) {
    const int lo = off[q];
    if (slot < off[q + 1] - lo) buf[(size_t)lo + slot] = c;
  } else if (slot < cap) {
    buf[(size_t)q * cap + slot] = c;
  }
}

// One wave = 64 queries (four 16-query tiles, their operands in registers), sweeping all point tiles: 4 x DT
// MFMAs per point tile, then 16 (query, point) pairs per lane are tested against the query's bound.
// F = double: v_mfma_f64_16x16x4f64.  F = float: v_mfma_f32_16x16x4f32 at twice the rate - the sweep only has to
// admit a SUPERSET of the true neighbours (k_bf_select decides in the KD-tree's arithmetic), so single precision
// does when its rounding margin, (4 DT + 8) 2^-24 (|x|^2 + |y|^2), is small against the bounds (the caller checks);
// the bound is rounded up and the point's norm down, towards admitting.
typedef float bf_f4 __attribute__((ext_vector_type(4)));
template <int DT, typename F>
__global__ __launch_bounds__(256) void k_bf_collect(const F* __restrict__ xop, const double* __restrict__ norm,
                                                    const double* __restrict__ tau, int N, int phase, int step,
                                                    int count, double margin_scale, double norm_max, int cap,
                                                    const int* __restrict__ off, int* __restrict__ cnt,
                                                    int* __restrict__ buf) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, cq = lane & 15;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q0 = wave * 64;
  if (q0 >= N) return;
  constexpr bool F32 = sizeof(F) == 4;
  typedef typename std::conditional<F32, bf_f4, bf_d4>::type acc_t;
  const int ntiles = (N + 15) / 16;
  F a[4][DT];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const long long tile = (long long)(q0 / 16 + qt);
#pragma unroll
    for (int t = 0; t < DT; ++t) a[qt][t] = tile < ntiles ? xop[(tile * DT + t) * 64 + lane] : F(0);
  }
  // this lane's 16 queries - rows kq + 4 e of a query tile in the fp64 product's result layout, rows 4 kq + e in
  // the fp32 one's -; bound minus the query's own norm (what is compared is |y|^2 - 2 x.y), with the rounding
  // margin of the product form
  auto query_of = [&](int qt, int e) { return q0 + 16 * qt + (F32 ? 4 * kq + e : kq + 4 * e); };
  F lim[4][4];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = query_of(qt, e);
      if (q < N) {
        const double nq = norm[q];
        const double l = tau[q] - nq + margin_scale * (nq + norm_max);
        if constexpr (F32) lim[qt][e] = __double2float_ru(l);
        else lim[qt][e] = l;
      } else {
        lim[qt][e] = F32 ? F(-3e38) : F(-1e300);
      }
    }
  // Software pipeline: a tile's operands AND its norms are loaded one tile ahead, so no load issued in an
  // iteration is waited for in that iteration (the first version loaded the norm at the top of the tile and
  // paid an L2 round trip per tile: the matrix cores idled two thirds of the time).  The four query tiles'
  // product chains are interleaved (four independent accumulators in flight, no stall on a chain's own result),
  // and the 16 (query, point) tests of a lane are folded into one mask: a single wave-wide branch per tile
  // guards the rare appends.  (The point itself is not filtered here: k_bf_select skips it.)
  __shared__ unsigned long long queue_all[4][128];  // (query << 32) | point
  volatile unsigned long long* queue = queue_all[threadIdx.x >> 6];
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int queued = 0;  // wave-uniform
  F b[DT], b_next[DT], n_next;
  auto norm_of = [&](int i) {  // (the padding points' 1e300 stays "infinitely far" in single precision)
    const double v = norm[i];
    if constexpr (F32) return v > 3e38 ? 3e38f : __double2float_rd(v);
    else return v;
  };
  // the j-th point tile of this sweep: phase 0 takes every step-th tile (a strided subset of the cloud: a fair
  // sample whatever order the points come in), phase 1 the tiles in between
  auto tile_of = [&](int j) { return phase == 0 ? j * step : (j / (step - 1)) * step + 1 + j % (step - 1); };
  int ct_next = tile_of(0);
  {
    n_next = norm_of(ct_next * 16 + cq);
#pragma unroll
    for (int t = 0; t < DT; ++t) b_next[t] = xop[((long long)ct_next * DT + t) * 64 + lane];
  }
  for (int j = 0; j < count; ++j) {
    const int ct = ct_next;
    const F nc = n_next;
#pragma unroll
    for (int t = 0; t < DT; ++t) b[t] = b_next[t];
    if (j + 1 < count) {
      ct_next = tile_of(j + 1);
      n_next = norm_of(ct_next * 16 + cq);  // (padded: no bounds test between the load and its use a tile later)
#pragma unroll
      for (int t = 0; t < DT; ++t) b_next[t] = xop[((long long)ct_next * DT + t) * 64 + lane];
    }
    acc_t acc[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) acc[qt] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        if constexpr (F32) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qt][t], b[t], acc[qt], 0, 0, 0);
        else acc[qt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qt][t], b[t], acc[qt], 0, 0, 0);
      }
    // D layout: lane (kq, cq) holds query rows kq + 4 e (fp64) / 4 kq + e (fp32), point column cq
    // (the tests are folded with a scalar OR of the compare masks: two vector instructions per pair)
    bool any = false;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int e = 0; e < 4; ++e) any |= nc - F(2) * acc[qt][e] <= lim[qt][e];
    if (__builtin_amdgcn_ballot_w64(any) != 0) {
      // Hits are parked in the wave's LDS queue and appended 64 at a time: an append is an atomic (whose return the
      // wave must wait for) plus a scattered store, and about 60 % of the tiles hit something - appending on the
      // spot stalled the wave a memory round trip per tile.
      const int c = ct * 16 + cq;
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool hit = nc - F(2) * acc[qt][e] <= lim[qt][e];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m != 0) {
            if (hit) {
              const int pos = queued + __popcll(m & lt_mask);
              queue[pos] = ((unsigned long long)(unsigned)query_of(qt, e) << 32) | (unsigned)c;
            }
            queued += __popcll(m);
            if (queued >= 64) {  // at most 127 parked: room for one more round of 64
              __builtin_amdgcn_wave_barrier();
              const unsigned long long it = queue[lane];
              const int iq = (int)(it >> 32), ic = (int)(unsigned)it;
              const int slot = atomicAdd(&cnt[iq], 1);
              bf_append(off, cap, iq, slot, ic, buf);
              const int rest = queued - 64;
              unsigned long long mv = 0;
              if (lane < rest) mv = queue[64 + lane];
              __builtin_amdgcn_wave_barrier();
              if (lane < rest) queue[lane] = mv;
              __builtin_amdgcn_wave_barrier();
              queued = rest;
            }
          }
        }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < queued) {  // what is still parked
    const unsigned long long it = queue[lane];
    const int iq = (int)(it >> 32), ic = (int)(unsigned)it;
    const int slot = atomicAdd(&cnt[iq], 1);
    bf_append(off, cap, iq, slot, ic, buf);
  }
}

// Between the two sweeps: the exact k-th smallest key among a query's candidates so far (all from the first
// point range) is a tighter upper bound of its k-th neighbour's key than the sample gave
template <int KMAX, int DT>
__global__ __launch_bounds__(128) void k_bf_refine(const double* __restrict__ x, int N, int d, int k, int cap,
                                                   const int* __restrict__ cnt, const int* __restrict__ buf,
                                                   double* __restrict__ tau) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  double q[4 * DT];
  bf_load_query<DT>(q, x + (size_t)i * d, d);
  double bd[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) bd[t] = 1e300;
  const int n = min(cnt[i], cap);
  for (int a = 0; a < n; ++a) {
    const int idx = buf[(size_t)i * cap + a];
    if (idx == i) continue;
    double cd = knn_sqdist_reg<DT>(q, x + (size_t)idx * d, d);
    if (cd < bd[KMAX - 1]) {
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const bool lt = cd < bd[t];
        const double td = bd[t];
        bd[t] = lt ? cd : td;
        cd = lt ? td : cd;
      }
    }
  }
  double kth = 1e300;
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t == k - 1) kth = bd[t];
  if (kth < tau[i]) tau[i] = kth;
}

// exact evaluation of the candidates (or of every point when the list overflowed / no list was made) and
// selection of the k smallest by (key, index); nearest first
template <int KMAX, int DT>
__global__ __launch_bounds__(128) void k_bf_select(const double* __restrict__ x, int N, int d, int k, int metric,
                                                   int cap, const int* __restrict__ cnt, const int* __restrict__ buf,
                                                   int* __restrict__ nn, double* __restrict__ dist,
                                                   int* __restrict__ n_scans) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  double q[4 * DT];
  bf_load_query<DT>(q, x + (size_t)i * d, d);
  double bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) {
    bd[t] = 1e300;
    bi[t] = 0x7fffffff;
  }
  const int have = cnt ? cnt[i] : -1;
  const bool scan = have < 0 || have > cap || have < k;
  const int n = scan ? N : have;
  if (scan && cnt) atomicAdd(n_scans, 1);
  for (int a = 0; a < n; ++a) {
    const int idx = scan ? a : buf[(size_t)i * cap + a];
    if (idx == i) continue;
    double cd = knn_key_reg<DT>(q, x + (size_t)idx * d, d, metric);
    int ci = idx;
    if (cd < bd[KMAX - 1] || (cd == bd[KMAX - 1] && ci < bi[KMAX - 1])) {
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const bool lt = cd < bd[t] || (cd == bd[t] && ci < bi[t]);
        const double td = bd[t];
        const int ti = bi[t];
        bd[t] = lt ? cd : td;
        bi[t] = lt ? ci : ti;
        cd = lt ? td : cd;
        ci = lt ? ti : ci;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t < k) {
      nn[(size_t)i * k + t] = bi[t];
      dist[(size_t)i * k + t] = metric == 0 ? knn_sqrt(bd[t]) : bd[t];
    }
}

}  // namespace gspx

template <int KMAX, int DT>
static void launch_bf_kd(gspx_ctx* ctx, const double* x, int N, int d, int k, int metric, int stride, double* tau,
                         int cap, const int* cnt, const int* buf, int* nn, double* dist, int* n_scans, int what) {
  const dim3 grid((unsigned)((N + 127) / 128));
  if (what == 0)
    hipLaunchKernelGGL((gspx::k_bf_tau<KMAX, DT>), grid, dim3(128), 0, ctx->stream, x, N, d, k, stride, tau);
  else if (what == 1)
    hipLaunchKernelGGL((gspx::k_bf_refine<KMAX, DT>), grid, dim3(128), 0, ctx->stream, x, N, d, k, cap, cnt, buf, tau);
  else
    hipLaunchKernelGGL((gspx::k_bf_select<KMAX, DT>), grid, dim3(128), 0, ctx->stream, x, N, d, k, metric, cap, cnt, buf,
                       nn, dist, n_scans);
}
template <int KMAX>
static void launch_bf_k(gspx_ctx* ctx, const double* x, int N, int d, int k, int metric, int stride, double* tau, int cap,
                        const int* cnt, const int* buf, int* nn, double* dist, int* n_scans, int what) {
  if (d <= 16) launch_bf_kd<KMAX, 4>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
  else if (d <= 32) launch_bf_kd<KMAX, 8>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
  else launch_bf_kd<KMAX, 16>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
}
// what: 0 the sample bounds (k_bf_tau), 1 their refinement between the sweeps (k_bf_refine), 2 the selection
static void launch_bf(gspx_ctx* ctx, const double* x, int N, int d, int k, int metric, int stride, double* tau, int cap,
                      const int* cnt, const int* buf, int* nn, double* dist, int* n_scans, int what) {
  if (k <= 8) launch_bf_k<8>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
  else if (k <= 16) launch_bf_k<16>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
  else if (k <= 32) launch_bf_k<32>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
  else launch_bf_k<64>(ctx, x, N, d, k, metric, stride, tau, cap, cnt, buf, nn, dist, n_scans, what);
}

// x: N x d doubles on the device; nn / dist: N x k outputs (nearest first, the point itself excluded).
// stats (nullable, 4 values): sample size, candidate capacity per query, mean candidates per query, queries that
// fell back to the exact scan
static int knn_bruteforce(gspx_ctx* ctx, const double* x, int N, int d, int k, int metric, int* nn, double* dist,
                          double* stats) {
  hipStream_t st = ctx->stream;
  DevMem n_scans;
  CHK(n_scans.alloc(64));
  HIPCHK(hipMemsetAsync(n_scans.p, 0, 64, st));
  if (metric != 0 || N <= 4 * k + 64) {  // no product form (or too few points to bother): exact scan for everybody
    launch_bf(ctx, x, N, d, k, metric, 1, nullptr, 0, nullptr, nullptr, nn, dist, n_scans.as<int>(), 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (stats) stats[0] = 0, stats[1] = 0, stats[2] = (double)N, stats[3] = (double)N;
    return GSPX_OK;
  }
  const int DT = d <= 16 ? 4 : (d <= 32 ? 8 : 16);
  const int ntiles = (N + 15) / 16;
  // sample: about max(2048, N / 128) points (at least 4 k + 1), strided through the cloud
  const int want = std::max(std::max(2048, N / 128), 4 * k + 1);
  const int stride = std::max(1, N / want);
  const int M = (N + stride - 1) / stride;
  // Two sweeps.  The sample's bound admits about k N / M points per query; sweeping only N1 points (every step-th
  // tile of the cloud) with it (k N1 / M candidates), tightening every bound to the exact k-th smallest of those (the k-th neighbour among
  // N1 points: admits about k N / N1 of all points), then sweeping the rest, appends k (N1 / M + N / N1) candidates
  // per query - least at N1 = sqrt(N M), a fifth of the single sweep's at N = 200k.  (The appends, one atomic and
  // one scattered store each, were what the single sweep spent its time on, not the products.)
  int step = (int)std::floor((double)ntiles / std::max(1.0, std::ceil(std::sqrt((double)N * (double)M) / 16.0)));
  if (step < 2) step = 1;                        // small clouds: one sweep over every tile
  const int t1 = (ntiles + step - 1) / step;     // tiles 0, step, 2 step, ...: the first sweep
  const double n1 = std::min<double>((double)t1 * 16.0, (double)N);
  const double expect = (double)k * (n1 / std::max(M - 1, 1) + (step > 1 ? (double)N / n1 : 0.0));
  // room for four times the expectation (clustered clouds), at least 8 k (+ 1: the point itself passes its own bound)
  const int cap = (int)std::min<int64_t>(std::max<int64_t>((int64_t)(4.0 * expect), 8 * k) + 1, N);
  DevMem xop, xop32, norm, tau, cnt, buf, pmax;
  CHK(xop.alloc((size_t)ntiles * DT * 64 * sizeof(double)));
  const bool try32 = ctx->opt.knn_f32 != 0;
  if (try32) CHK(xop32.alloc((size_t)ntiles * DT * 64 * sizeof(float)));
  CHK(norm.alloc((size_t)ntiles * 16 * sizeof(double)));
  CHK(tau.alloc((size_t)N * sizeof(double)));
  CHK(cnt.alloc((size_t)N * sizeof(int)));
  CHK(buf.alloc((size_t)N * cap * sizeof(int)));
  HIPCHK(hipMemsetAsync(cnt.p, 0, (size_t)N * sizeof(int), st));
  const long long total = (long long)ntiles * DT * 64;
  hipLaunchKernelGGL(gspx::k_bf_prepare, dim3((unsigned)((std::max<long long>(total, N) + 255) / 256)), dim3(256), 0, st, x,
                     N, d, DT, xop.as<double>(), norm.as<double>(), try32 ? xop32.as<float>() : (float*)nullptr);
  launch_bf(ctx, x, N, d, k, metric, stride, tau.as<double>(), 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<double> hn((size_t)N), ht((size_t)N);
  HIPCHK(hipMemcpyAsync(hn.data(), norm.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (try32) HIPCHK(hipMemcpyAsync(ht.data(), tau.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  double nmax = 0;
  for (double v : hn) nmax = std::max(nmax, v);
  // |fl(|y|^2 - 2 x.y) + |x|^2 - |x - y|^2| <= (4 DT + 8) u (|x|^2 + |y|^2) with u = 2^-53; ten times that
  const double margin64 = 10.0 * (4.0 * DT + 8.0) * 1.1102230246251565e-16;
  // The same bound with u = 2^-24 for the single-precision sweep: used when it widens a typical bound (the
  // median over a sample of the queries) by at most half a percent - clouds far from the origin (|x|^2 >> the
  // neighbour distances) keep the fp64 sweep.
  const double margin32 = 10.0 * (4.0 * DT + 8.0) * 5.9604644775390625e-08;
  bool use32 = false;
  if (try32 && nmax < 1e30) {
    std::vector<double> sample;
    for (size_t i = 0; i < (size_t)N; i += std::max<size_t>(1, (size_t)N / 2048)) sample.push_back(ht[i]);
    std::nth_element(sample.begin(), sample.begin() + sample.size() / 2, sample.end());
    const double tmed = sample[sample.size() / 2];
    use32 = ctx->opt.knn_f32 == 2 || margin32 * 2.0 * nmax <= 0.005 * tmed;
  }
  const double margin = use32 ? margin32 : margin64;
  const dim3 grid((unsigned)((N + 255) / 256));
#define GSPX_BF(D_, PH_, CNT_)                                                                                           \
  do {                                                                                                                   \
    if (use32)                                                                                                           \
      hipLaunchKernelGGL((gspx::k_bf_collect<D_, float>), grid, dim3(256), 0, st, xop32.as<float>(), norm.as<double>(),  \
                         tau.as<double>(), N, PH_, step, CNT_, margin, nmax, cap, (const int*)nullptr, cnt.as<int>(),    \
                         buf.as<int>());                                                                                 \
    else                                                                                                                 \
      hipLaunchKernelGGL((gspx::k_bf_collect<D_, double>), grid, dim3(256), 0, st, xop.as<double>(), norm.as<double>(),  \
                         tau.as<double>(), N, PH_, step, CNT_, margin, nmax, cap, (const int*)nullptr, cnt.as<int>(),    \
                         buf.as<int>());                                                                                 \
  } while (0)
  auto sweep = [&](int phase, int count) {
    if (DT == 4) GSPX_BF(4, phase, count);
    else if (DT == 8) GSPX_BF(8, phase, count);
    else GSPX_BF(16, phase, count);
  };
  sweep(0, t1);
  if (step > 1) {
    launch_bf(ctx, x, N, d, k, metric, stride, tau.as<double>(), cap, cnt.as<int>(), buf.as<int>(), nullptr, nullptr, nullptr, 1);
    sweep(1, ntiles - t1);
  }
#undef GSPX_BF
  launch_bf(ctx, x, N, d, k, metric, stride, nullptr, cap, cnt.as<int>(), buf.as<int>(), nn, dist, n_scans.as<int>(), 2);
  HIPCHK(hipGetLastError());
  int scans = 0;
  HIPCHK(hipMemcpyAsync(&scans, n_scans.p, sizeof(int), hipMemcpyDeviceToHost, st));
  std::vector<int> hc;
  if (stats) {
    hc.resize((size_t)N);
    HIPCHK(hipMemcpyAsync(hc.data(), cnt.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  if (stats) {
    double s = 0;
    for (int v : hc) s += v;
    stats[0] = M;
    stats[1] = cap;
    stats[2] = s / N;
    stats[3] = scans;
  }
  return GSPX_OK;
}

// ---- radius search in more than three dimensions (NNtype='radius', nngraph.py:228-287) -------------------------
namespace gspx {

// PASS 0: count the points within the radius (key <= eps2, the KD-tree's ball-query criterion in its own
// arithmetic), PASS 1: write them at rowptr[i].  Candidates from the MFMA sweep (off / buf), or every point.
template <int PASS, int DT>
__global__ __launch_bounds__(128) void k_bf_radius(const double* __restrict__ x, int N, int d, int metric, double eps2,
                                                   const int* __restrict__ off, const int* __restrict__ buf,
                                                   int* __restrict__ cnt, const int* __restrict__ rowptr,
                                                   int* __restrict__ col, double* __restrict__ dist) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  double q[4 * DT];
  bf_load_query<DT>(q, x + (size_t)i * d, d);
  const int lo = off ? off[i] : 0, n = off ? off[i + 1] - lo : N;
  const int o = PASS ? rowptr[i] : 0;
  int m = 0;
  for (int a = 0; a < n; ++a) {
    const int idx = off ? buf[(size_t)lo + a] : a;
    if (idx == i) continue;
    const double key = knn_key_reg<DT>(q, x + (size_t)idx * d, d, metric);
    if (key <= eps2) {
      if (PASS) {
        col[o + m] = idx;
        dist[o + m] = metric == 0 ? knn_sqrt(key) : key;
      }
      ++m;
    }
  }
  if (!PASS) cnt[i] = m;
}

}  // namespace gspx

template <int PASS>
static void launch_bf_radius(gspx_ctx* ctx, const double* x, int N, int d, int metric, double eps2, const int* off,
                             const int* buf, int* cnt, const int* rowptr, int* col, double* dist) {
  const dim3 grid((unsigned)((N + 127) / 128));
#define GSPX_BR(D_)                                                                                                   \
  hipLaunchKernelGGL((gspx::k_bf_radius<PASS, D_>), grid, dim3(128), 0, ctx->stream, x, N, d, metric, eps2, off, buf, cnt, \
                     rowptr, col, dist)
  if (d <= 16) GSPX_BR(4);
  else if (d <= 32) GSPX_BR(8);
  else GSPX_BR(16);
#undef GSPX_BR
}

// Candidate lists of the radius search: per query every point whose product-form squared distance is within
// eps2 (+ the rounding margin).  Two sweeps with the same bound: one counts, one fills rows of exactly those lengths.
// Leaves off (N + 1) and buf; euclidean only (the other metrics scan every point in k_bf_radius).
static int radius_candidates(gspx_ctx* ctx, const double* x, int N, int d, double eps2, DevMem& off, DevMem& buf) {
  hipStream_t st = ctx->stream;
  const int DT = d <= 16 ? 4 : (d <= 32 ? 8 : 16);
  const int ntiles = (N + 15) / 16;
  DevMem xop, norm, tau, cnt;
  CHK(xop.alloc((size_t)ntiles * DT * 64 * sizeof(double)));
  CHK(norm.alloc((size_t)ntiles * 16 * sizeof(double)));
  CHK(tau.alloc((size_t)N * sizeof(double)));
  CHK(cnt.alloc(((size_t)N + 1) * sizeof(int)));
  CHK(off.alloc(((size_t)N + 1) * sizeof(int)));
  HIPCHK(hipMemsetAsync(cnt.p, 0, ((size_t)N + 1) * sizeof(int), st));
  const long long total = (long long)ntiles * DT * 64;
  hipLaunchKernelGGL(gspx::k_bf_prepare, dim3((unsigned)((std::max<long long>(total, N) + 255) / 256)), dim3(256), 0, st, x,
                     N, d, DT, xop.as<double>(), norm.as<double>(), (float*)nullptr);
  hipLaunchKernelGGL((k_fill<double>), dim3(1024), dim3(256), 0, st, tau.as<double>(), (size_t)N, eps2);
  std::vector<double> hn((size_t)N);
  HIPCHK(hipMemcpyAsync(hn.data(), norm.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  double nmax = 0;
  for (double v : hn) nmax = std::max(nmax, v);
  const double margin = 10.0 * (4.0 * DT + 8.0) * 1.1102230246251565e-16;
  const dim3 grid((unsigned)((N + 255) / 256));
  auto sweep = [&](const int* offp, int* bufp) {
#define GSPX_BF(D_)                                                                                                  \
  hipLaunchKernelGGL((gspx::k_bf_collect<D_, double>), grid, dim3(256), 0, st, xop.as<double>(), norm.as<double>(),  \
                     tau.as<double>(), N, 0, 1, ntiles, margin, nmax, 0, offp, cnt.as<int>(), bufp)
    if (DT == 4) GSPX_BF(4);
    else if (DT == 8) GSPX_BF(8);
    else GSPX_BF(16);
#undef GSPX_BF
  };
  sweep(nullptr, nullptr);  // count only
  {  // the total in 64 bits BEFORE the 32-bit scan: a sum beyond 2^32 would wrap back to a plausible positive value
    std::vector<int> hc((size_t)N);
    HIPCHK(hipMemcpyAsync(hc.data(), cnt.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t total64 = 0;
    for (int v : hc) total64 += v;
    if (total64 >= ((int64_t)1 << 31))
      return set_err(GSPX_ERR_INVALID, "gspx_radius_build: %lld candidate pairs, more than 2^31 (epsilon too large)",
                     (long long)total64);
  }
  CHK(scan_exclusive(ctx, cnt.as<int>(), off.as<int>(), N + 1));
  int total_c = 0;
  HIPCHK(hipMemcpyAsync(&total_c, off.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (total_c < 0) return set_err(GSPX_ERR_INVALID, "gspx_radius_build: more than 2^31 candidate pairs (epsilon too large)");
  CHK(buf.alloc((size_t)std::max(total_c, 1) * sizeof(int)));
  HIPCHK(hipMemsetAsync(cnt.p, 0, ((size_t)N + 1) * sizeof(int), st));
  sweep(off.as<int>(), buf.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return GSPX_OK;
}
