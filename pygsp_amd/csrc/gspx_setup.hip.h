// gspx_setup.hip.h - graph set-up in ONE call, on the device.  Included by gspx.hip.
//
// What pygsp/graphs/graph.py:98-176 + 510-630 do to a weight matrix before the first filter call - the checks
// of Graph.__init__ (NaN / inf, self-loops, negative weights, graph.py:108-134), the directedness test
// (`(W - W.T)` of Graph.is_directed, graph.py:357-405), the Laplacian (graph.py:510-630) - plus this engine's
// own preparation (canonical-CSR validation, the space-filling-curve order of the vertices, its locality
// score) used to be host numpy / scipy passes over the stored entries: 0.27 s for the 1M-vertex headline graph
// against 2.9 ms of device build, i.e. one graph cost 25 filter calls.  Here W is uploaded once and everything
// runs on it in place:
//   k_w_inspect     one pass over the stored entries: CSR validity (row bounds, ascending columns, range), NaN /
//                   inf / negative / explicit-zero / diagonal counts, and the symmetry test - for every stored
//                   (i, j, v) the mirror (j, i) is looked up by binary search in row j and compared bit for bit
//   k_setup_convert the values in the compute dtype (int64 adjacency of ER / SBM graphs -> float, as scipy does)
//   k_curve_keys + radix sort (k_radix_hist / scan / k_radix_scatter, 8 bits per pass, stable): the vertex order
//                   along the Hilbert / Morton curve = numpy's stable argsort of the keys
//   k_locality      share of stored entries whose two vertices are at most `reach` positions apart, in the
//                   graph's own order and in the curve order (the order is kept only if it wins by 0.05)
// and then the existing device build (degrees, Laplacian, internal layout).  A directed graph - the reference builds
// its Laplacian from utils.symmetrize(W, 'average') = (W + W.T) / 2, graph.py:613-616, utils.py:247-248 - and a W with
// explicit zeros to drop are prepared on the device too (round 5): transpose by the same stable radix sort (keys = the
// column indices), a two-pointer merge of row i of W with row i of its transpose, entries whose sum is zero dropped
// as scipy's sparse addition drops them (k_sym_count / k_sym_fill).
#pragma once

namespace gspx {

struct WReport {  // device-side counters
  unsigned long long nan, inf, neg, zero, diag, asym, bad, differ;
};

template <typename TIn> __device__ __forceinline__ bool setup_isnan(TIn v) { return v != v; }
template <> __device__ __forceinline__ bool setup_isnan<long long>(long long) { return false; }
template <typename TIn> __device__ __forceinline__ bool setup_isinf(TIn v) { return isinf((double)v); }
template <> __device__ __forceinline__ bool setup_isinf<long long>(long long) { return false; }

template <typename TIn>
__global__ __launch_bounds__(256) void k_w_inspect(const int* __restrict__ ptr, const int* __restrict__ col,
                                                   const TIn* __restrict__ val, int N, long long nnz,
                                                   WReport* __restrict__ rep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned nan = 0, inf = 0, neg = 0, zero = 0, diag = 0, asym = 0, bad = 0, differ = 0;
  if (i < N) {
    const long long s = ptr[i], e = ptr[i + 1];
    if (s < 0 || e < s || e > nnz || (i == 0 && s != 0) || (i == N - 1 && e != nnz)) {
      bad = 1;
    } else {
      int prev = -1;
      for (long long j = s; j < e; ++j) {
        const int c = col[j];
        const TIn v = val[j];
        if (c < 0 || c >= N || c <= prev) {
          ++bad;
          prev = c;
          continue;
        }
        prev = c;
        nan += setup_isnan(v);
        inf += setup_isinf(v);
        neg += v < TIn(0);
        zero += v == TIn(0);
        if (c == i) {
          diag += v != TIn(0);
          continue;
        }
        // the mirror entry (c, i): binary search in row c (its bounds are checked before any read)
        const long long cs = ptr[c], ce = ptr[c + 1];
        bool same = false;
        TIn mirror = TIn(0);  // a mirror entry that is not stored is a zero
        if (cs >= 0 && ce >= cs && ce <= nnz) {
          long long lo = cs, hi = ce;
          while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (col[mid] < i) lo = mid + 1;
            else hi = mid;
          }
          const bool found = lo < ce && col[lo] == i;
          if (found) mirror = val[lo];
          same = found && (val[lo] == v || (setup_isnan(v) && setup_isnan(val[lo])));
        }
        asym += !same;                // the fast path needs every mirror STORED and equal bit for bit ...
        differ += !(v == mirror);     // ... Graph.is_directed (graph.py:357-405) asks whether W - W.T has a nonzero
      }
    }
  }
  // one atomic per counter and wave that has something to report
  auto flush = [](unsigned v, unsigned long long* dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, (unsigned long long)v);
  };
  flush(nan, &rep->nan);
  flush(inf, &rep->inf);
  flush(neg, &rep->neg);
  flush(zero, &rep->zero);
  flush(diag, &rep->diag);
  flush(asym, &rep->asym);
  flush(bad, &rep->bad);
  flush(differ, &rep->differ);
}

template <typename TIn, typename T>
__global__ void k_setup_convert(const TIn* __restrict__ in, size_t n, T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (T)in[i];
}

// ---- (W + W.T) / 2 of a canonical CSR matrix, zero sums dropped (utils.symmetrize 'average' on scipy matrices) ------
// The value type of the sum is the input's (float: float; double and int64: double - an int64 sum is exact, scipy
// divides it in float64), so the result equals scipy's bit for bit.
template <typename TIn> struct SymType { typedef double t; };
template <> struct SymType<float> { typedef float t; };
template <typename TIn> __device__ __forceinline__ typename SymType<TIn>::t sym_half_sum(TIn a, TIn b) {
  typedef typename SymType<TIn>::t TS;
  return (TS)(a + b) * TS(0.5);  // (long long: the sum first, exactly; then one conversion)
}

__global__ void k_sym_keys(const int* __restrict__ ptr, const int* __restrict__ col, int N,
                           unsigned long long* __restrict__ keys, int* __restrict__ rows, int* __restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
    const int c = col[j];
    keys[j] = (unsigned long long)(unsigned)c;  // sorted by column, stable: by (column, row) = the rows of W.T in order
    rows[j] = i;
    atomicAdd(&cnt[c], 1);
  }
}

// Row i of W (ascending columns) merged with row i of W.T (entries order[t], t in [tptr[i], tptr[i+1]): ascending
// original rows).  FILL == false: the number of entries with a nonzero sum -> cnt[i]; true: the entries -> out.
template <typename TIn, bool FILL>
__global__ __launch_bounds__(256) void k_sym_merge(const int* __restrict__ ptr, const int* __restrict__ col,
                                                   const TIn* __restrict__ val, const int* __restrict__ tptr,
                                                   const int* __restrict__ order, const int* __restrict__ rows, int N,
                                                   int* __restrict__ cnt, const int* __restrict__ optr,
                                                   int* __restrict__ ocol, typename SymType<TIn>::t* __restrict__ oval) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int a = ptr[i], t = tptr[i];
  const int ae = ptr[i + 1], te = tptr[i + 1];
  int n = 0, o = FILL ? optr[i] : 0;
  while (a < ae || t < te) {
    const int ca = a < ae ? col[a] : 0x7FFFFFFF;
    const int e = t < te ? order[t] : 0;
    const int ct = t < te ? rows[e] : 0x7FFFFFFF;
    const int c = ca < ct ? ca : ct;
    const TIn va = ca == c ? val[a] : TIn(0), vt = ct == c ? val[e] : TIn(0);
    a += ca == c;
    t += ct == c;
    if ((va + vt) != TIn(0)) {  // scipy's sparse sum keeps no zero (csr_binop_csr)
      if (FILL) {
        ocol[o] = c;
        oval[o] = sym_half_sum<TIn>(va, vt);
        ++o;
      }
      ++n;
    }
  }
  if (!FILL) cnt[i] = n;
}

// ---- stable LSD radix sort of (64-bit key, 32-bit payload) pairs, 8 bits per pass -----------------------
// One 64-lane workgroup owns a tile of RADIX_TILE consecutive elements.  Pass 1 counts the digits of its tile
// (LDS atomics) into hist[digit][tile]; an exclusive scan of that table in digit-major order gives every
// (digit, tile) its first output slot; pass 2 walks the tile again in order, 64 elements at a time: lanes
// holding the same digit find each other with eight ballots (one per digit bit), their rank among themselves is
// a popcount of the lower lanes - so equal digits keep their order, which is what makes the sort stable.
#define GSPX_RADIX_TILE 2048

__global__ __launch_bounds__(64) void k_radix_hist(const unsigned long long* __restrict__ key, int n, int shift,
                                                   int ntiles, int* __restrict__ hist) {
  __shared__ int h[256];
  const int lane = threadIdx.x, tile = blockIdx.x;
  for (int d = lane; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const int base = tile * GSPX_RADIX_TILE;
  for (int k = lane; k < GSPX_RADIX_TILE && base + k < n; k += 64)
    atomicAdd(&h[(int)((key[base + k] >> shift) & 255ull)], 1);
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(size_t)d * ntiles + tile] = h[d];
}

__global__ __launch_bounds__(64) void k_radix_scatter(const unsigned long long* __restrict__ key,
                                                      const int* __restrict__ val, int n, int shift, int ntiles,
                                                      const int* __restrict__ offs,
                                                      unsigned long long* __restrict__ key_out,
                                                      int* __restrict__ val_out) {
  __shared__ int next[256];
  const int lane = threadIdx.x, tile = blockIdx.x;
  for (int d = lane; d < 256; d += 64) next[d] = offs[(size_t)d * ntiles + tile];
  __syncthreads();
  const int base = tile * GSPX_RADIX_TILE;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int k0 = 0; k0 < GSPX_RADIX_TILE && base + k0 < n; k0 += 64) {
    const int idx = base + k0 + lane;
    const bool live = idx < n;
    const unsigned long long kv = live ? key[idx] : 0ull;
    const int pv = live ? (val ? val[idx] : idx) : 0;
    const int d = (int)((kv >> shift) & 255ull);
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot(live && ((d >> b) & 1));
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int first = live ? next[d] : 0;  // every lane reads before any lane of the wave writes
    __builtin_amdgcn_wave_barrier();
    if (live) {
      const int rank = __popcll(peers & lt);
      key_out[first + rank] = kv;
      val_out[first + rank] = pv;
      if ((peers >> lane) == 1ull) next[d] = first + __popcll(peers);  // the highest peer lane moves the counter
    }
    __syncthreads();
  }
}

__global__ void k_locality(const int* __restrict__ ptr, const int* __restrict__ col, int N, const int* __restrict__ inv,
                           int reach, unsigned long long* __restrict__ out /* own, curve */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned own = 0, cur = 0;
  if (i < N) {
    const int pi = inv ? inv[i] : i;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
      const int c = col[j];
      own += abs(i - c) <= reach;
      if (inv) cur += abs(pi - inv[c]) <= reach;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    own += __shfl_down(own, off);
    cur += __shfl_down(cur, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (own) atomicAdd(&out[0], (unsigned long long)own);
    if (cur) atomicAdd(&out[1], (unsigned long long)cur);
  }
}

}  // namespace gspx

// keys[n] (device) -> order[n] (device): the stable argsort of the keys; `bits` significant key bits
static int radix_argsort(gspx_ctx* ctx, const unsigned long long* keys, int n, int bits, int* order) {
  if (n <= 0) return GSPX_OK;
  hipStream_t st = ctx->stream;
  const int ntiles = (n + GSPX_RADIX_TILE - 1) / GSPX_RADIX_TILE;
  DevMem k2, k3, v2, hist;
  CHK(k2.alloc((size_t)n * 8));
  CHK(k3.alloc((size_t)n * 8));
  CHK(v2.alloc((size_t)n * 4));
  CHK(hist.alloc(((size_t)256 * ntiles + 1) * sizeof(int)));
  const int passes = std::max(1, (bits + 7) / 8);
  // ping-pong: keys -> k2 -> k3 -> k2 ...; payload: (identity) -> a -> b -> a ..., the last pass lands in `order`
  const unsigned long long* kin = keys;
  const int* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    unsigned long long* kout = (p & 1) ? k3.as<unsigned long long>() : k2.as<unsigned long long>();
    // payload buffers alternate between v2 and order so that the final pass writes `order`
    int* vout = ((passes - 1 - p) & 1) ? v2.as<int>() : order;
    hipLaunchKernelGGL(gspx::k_radix_hist, dim3(ntiles), dim3(64), 0, st, kin, n, 8 * p, ntiles, hist.as<int>());
    CHK(scan_exclusive(ctx, hist.as<int>(), hist.as<int>(), 256 * ntiles));
    hipLaunchKernelGGL(gspx::k_radix_scatter, dim3(ntiles), dim3(64), 0, st, kin, vin, n, 8 * p, ntiles, hist.as<int>(),
                       kout, vout);
    HIPCHK(hipGetLastError());
    kin = kout;
    vin = vout;
  }
  HIPCHK(hipStreamSynchronize(st));
  return GSPX_OK;
}

extern "C" int gspx_curve_order(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve, int32_t* perm) {
  if (!ctx || !coords || !perm) return set_err(GSPX_ERR_INVALID, "null argument");
  if (N < 1 || N >= ((int64_t)1 << 31) || d < 2) return set_err(GSPX_ERR_INVALID, "gspx_curve_order: bad N or d");
  if (curve != 0 && curve != 1) return set_err(GSPX_ERR_INVALID, "curve: 0 Morton, 1 Hilbert");
  HIPCHK(hipSetDevice(ctx->device));
  DevMem keys, order;
  CHK(keys.alloc((size_t)N * 8));
  CHK(order.alloc((size_t)N * 4));
  CHK(curve_keys_dev(ctx, N, d, coords, curve, (unsigned long long*)keys.p));
  const int bits = curve == 1 ? 32 : (d >= 3 ? 63 : 62);
  CHK(radix_argsort(ctx, (const unsigned long long*)keys.p, (int)N, bits, order.as<int>()));
  HIPCHK(hipMemcpy(perm, order.p, (size_t)N * 4, hipMemcpyDeviceToHost));
  return GSPX_OK;
}

#define GSPX_I64 2 /* data_dtype of gspx_graph_setup only: int64 adjacency (ER / SBM graphs of the reference) */

// (W + W.T) / 2 without stored zeros, on the device: sptr / scol / sval receive a canonical CSR matrix
template <typename TIn>
static int symmetrize_dev(gspx_ctx* ctx, int N, int64_t nnz, const int* wptr, const int* wcol, const TIn* wval,
                          DevMem& sptr, DevMem& scol, DevMem& sval, int64_t* snnz) {
  typedef typename gspx::SymType<TIn>::t TS;
  hipStream_t st = ctx->stream;
  const int nb = std::max(1, (N + 255) / 256);
  DevMem keys, rows, order, tptr;
  CHK(keys.alloc((size_t)std::max<int64_t>(nnz, 1) * 8));
  CHK(rows.alloc((size_t)std::max<int64_t>(nnz, 1) * 4));
  CHK(order.alloc((size_t)std::max<int64_t>(nnz, 1) * 4));
  CHK(tptr.alloc((size_t)(N + 1) * 4));
  CHK(sptr.alloc((size_t)(N + 1) * 4));
  HIPCHK(hipMemsetAsync(tptr.p, 0, (size_t)(N + 1) * 4, st));
  HIPCHK(hipMemsetAsync(sptr.p, 0, (size_t)(N + 1) * 4, st));
  hipLaunchKernelGGL(gspx::k_sym_keys, dim3(nb), dim3(256), 0, st, wptr, wcol, N, (unsigned long long*)keys.p,
                     rows.as<int>(), tptr.as<int>());
  HIPCHK(hipGetLastError());
  CHK(scan_exclusive(ctx, tptr.as<int>(), tptr.as<int>(), N + 1));  // column counts -> row starts of W.T
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < (int64_t)N) ++bits;
  CHK(radix_argsort(ctx, (const unsigned long long*)keys.p, (int)nnz, bits, order.as<int>()));
  hipLaunchKernelGGL((gspx::k_sym_merge<TIn, false>), dim3(nb), dim3(256), 0, st, wptr, wcol, wval, tptr.as<int>(),
                     order.as<int>(), rows.as<int>(), N, sptr.as<int>(), (const int*)nullptr, (int*)nullptr, (TS*)nullptr);
  HIPCHK(hipGetLastError());
  CHK(scan_exclusive(ctx, sptr.as<int>(), sptr.as<int>(), N + 1));
  int total = 0;
  HIPCHK(hipMemcpy(&total, sptr.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost));
  *snnz = total;
  CHK(scol.alloc((size_t)std::max(total, 1) * 4));
  CHK(sval.alloc((size_t)std::max(total, 1) * sizeof(TS)));
  hipLaunchKernelGGL((gspx::k_sym_merge<TIn, true>), dim3(nb), dim3(256), 0, st, wptr, wcol, wval, tptr.as<int>(),
                     order.as<int>(), rows.as<int>(), N, (int*)nullptr, sptr.as<int>(), scol.as<int>(), sval.as<TS>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return GSPX_OK;
}

// W already on the device (uploaded by graph_setup_t below, or left there by the neighbour / block-model builders)
template <typename TIn, typename T>
static int graph_setup_core(gspx_graph* g, int64_t nnz, const int* wptr_d, const int* wcol_d, const TIn* wraw_d,
                            const double* coords, int d, int order_mode, const int32_t* perm_in, int64_t* report,
                            std::chrono::steady_clock::time_point t0);

template <typename TIn, typename T>
static int graph_setup_t(gspx_graph* g, int64_t nnz, const int32_t* indptr, const int32_t* indices, const void* data,
                         const double* coords, int d, int order_mode, const int32_t* perm_in, int64_t* report) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int N = (int)g->N;
  const auto t0 = std::chrono::steady_clock::now();
  DevMem wptr, wcol, wraw;
  CHK(wptr.alloc((size_t)(N + 1) * sizeof(int)));
  CHK(wcol.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
  CHK(wraw.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(TIn)));
  HIPCHK(hipMemcpyAsync(wptr.p, indptr, (size_t)(N + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  if (nnz > 0) {
    HIPCHK(hipMemcpyAsync(wcol.p, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(wraw.p, data, (size_t)nnz * sizeof(TIn), hipMemcpyHostToDevice, st));
  }
  return graph_setup_core<TIn, T>(g, nnz, wptr.as<int>(), wcol.as<int>(), (const TIn*)wraw.p, coords, d, order_mode,
                                  perm_in, report, t0);
}

template <typename TIn, typename T>
static int graph_setup_core(gspx_graph* g, int64_t nnz, const int* wptr_d, const int* wcol_d, const TIn* wraw_d,
                            const double* coords, int d, int order_mode, const int32_t* perm_in, int64_t* report,
                            std::chrono::steady_clock::time_point t0) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int N = (int)g->N;
  DevMem wval, rep;
  CHK(rep.alloc(sizeof(gspx::WReport) + 16));
  HIPCHK(hipMemsetAsync(rep.p, 0, sizeof(gspx::WReport) + 16, st));
  const int nb = std::max(1, (N + 255) / 256);
  if (N > 0)
    hipLaunchKernelGGL((gspx::k_w_inspect<TIn>), dim3(nb), dim3(256), 0, st, wptr_d, wcol_d, wraw_d, N, (long long)nnz,
                       (gspx::WReport*)rep.p);
  gspx::WReport r;
  HIPCHK(hipMemcpyAsync(&r, rep.p, sizeof(r), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  report[0] = (int64_t)r.nan;
  report[1] = (int64_t)r.inf;
  report[2] = (int64_t)r.neg;
  report[3] = (int64_t)r.zero;
  report[4] = (int64_t)r.diag;
  report[5] = (int64_t)r.differ;  // > 0: W - W.T has a nonzero = the graph is directed (graph.py:357-405)
  report[6] = (int64_t)r.bad;
  report[10] = 1;  // not built (yet)
  if (r.bad) return set_err(GSPX_ERR_INVALID, "W is not a canonical CSR matrix (row bounds, ascending column indices, range)");
  if (r.nan) return set_err(GSPX_ERR_INVALID, "Adjacency: there is a Not a Number (NaN).");  // graph.py:112-114
  if (r.inf) return set_err(GSPX_ERR_INVALID, "Adjacency: there is an infinite value.");       // graph.py:115-117
  // A directed graph (some mirror entry missing or different) or explicit zeros: the Laplacian is built from
  // (W + W.T) / 2 without stored zeros (graph.py:613-616 / :126-128), prepared here on the device.
  typedef typename gspx::SymType<TIn>::t TS;
  DevMem sptr, scol, sval;
  const bool prepared = r.asym || r.zero;
  if (prepared) {
    if (2 * nnz >= ((int64_t)1 << 31) - 8 * (int64_t)N - 64) return GSPX_OK;  // (its union pattern may not fit int32: host route)
    int64_t snnz = 0;
    CHK(symmetrize_dev<TIn>(ctx, N, nnz, wptr_d, wcol_d, wraw_d, sptr, scol, sval, &snnz));
    wptr_d = sptr.as<int>();
    wcol_d = scol.as<int>();
    nnz = snnz;
  }

  // ---- internal vertex order -----------------------------------------------------------------------------
  const bool want_curve = coords && d >= 2 && N >= 2 && (order_mode == 1 || order_mode == 2 || order_mode == 3);
  if (order_mode == 4 && perm_in) {
    CHK(upload_perm(g, perm_in));
  } else if (want_curve) {
    const int curve = order_mode == 3 || (order_mode == 1 && d == 2) ? 1 : 0;
    DevMem keys;
    CHK(keys.alloc((size_t)N * 8));
    CHK(g->perm.alloc((size_t)N * sizeof(int)));
    CHK(g->iperm.alloc((size_t)N * sizeof(int)));
    CHK(curve_keys_dev(ctx, N, d, coords, curve, (unsigned long long*)keys.p));
    CHK(radix_argsort(ctx, (const unsigned long long*)keys.p, N, curve == 1 ? 32 : (d >= 3 ? 63 : 62),
                      g->perm.as<int>()));
    hipLaunchKernelGGL(k_inverse_perm, dim3(nb), dim3(256), 0, st, g->perm.as<int>(), N, g->iperm.as<int>());
    g->has_perm = true;
    if (order_mode == 1) {  // "auto": kept only if it beats the graph's own order (engine.auto_order)
      const int reach = (int)std::min<int64_t>(8192, std::max<int64_t>(64, N / 64));
      unsigned long long* sc = (unsigned long long*)((char*)rep.p + sizeof(gspx::WReport));
      hipLaunchKernelGGL(gspx::k_locality, dim3(nb), dim3(256), 0, st, wptr_d, wcol_d, N, g->iperm.as<int>(), reach, sc);
      unsigned long long h[2] = {0, 0};
      HIPCHK(hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      const double den = nnz > 0 ? (double)nnz : 1.0;
      const double own = nnz > 0 ? (double)h[0] / den : 1.0, cur = nnz > 0 ? (double)h[1] / den : 1.0;
      report[8] = (int64_t)(own * 1e9);
      report[9] = (int64_t)(cur * 1e9);
      if (cur < own + 0.05) {
        g->has_perm = false;
        g->perm.release();
        g->iperm.release();
      }
    }
  }
  report[7] = g->has_perm ? 1 : 0;

  // ---- values in the compute dtype, then the ordinary device build ----------------------------------------
  const T* vals = nullptr;
  if (prepared) {  // the symmetrised values (float or double) in the compute dtype
    if (std::is_same<TS, T>::value) {
      vals = (const T*)sval.p;
    } else {
      CHK(wval.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(T)));
      if (nnz > 0)
        hipLaunchKernelGGL((gspx::k_setup_convert<TS, T>), dim3(2048), dim3(256), 0, st, sval.as<TS>(), (size_t)nnz, wval.as<T>());
      vals = wval.as<T>();
    }
  } else if (std::is_same<TIn, T>::value) {
    vals = (const T*)wraw_d;
  } else {
    CHK(wval.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(T)));
    if (nnz > 0)
      hipLaunchKernelGGL((gspx::k_setup_convert<TIn, T>), dim3(2048), dim3(256), 0, st, wraw_d, (size_t)nnz, wval.as<T>());
    vals = wval.as<T>();
  }
  CHK(create_from_w_dev<T>(g, nnz, wptr_d, wcol_d, vals));
  report[10] = 0;
  report[11] = (int64_t)(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  g->build_ms = (double)report[11] / 1e3;
  return GSPX_OK;
}

extern "C" int gspx_graph_setup(gspx_ctx* ctx, int64_t N, int64_t nnz, const int32_t* indptr, const int32_t* indices,
                                const void* data, int data_dtype, int lap_type, int compute_dtype, const double* coords,
                                int d, int order_mode, const int32_t* perm_in, int64_t report[12], gspx_graph** out) {
  if (!ctx || !out || !report) return set_err(GSPX_ERR_INVALID, "null ctx, report or output");
  *out = nullptr;
  for (int i = 0; i < 12; ++i) report[i] = 0;
  if (data_dtype != GSPX_F32 && data_dtype != GSPX_F64 && data_dtype != GSPX_I64)
    return set_err(GSPX_ERR_INVALID, "data_dtype must be GSPX_F32, GSPX_F64 or 2 (int64)");
  if (compute_dtype != GSPX_F32 && compute_dtype != GSPX_F64)
    return set_err(GSPX_ERR_INVALID, "compute_dtype must be GSPX_F32 or GSPX_F64");
  if (lap_type != GSPX_LAP_COMBINATORIAL && lap_type != GSPX_LAP_NORMALIZED)
    return set_err(GSPX_ERR_INVALID, "Unknown Laplacian type %d", lap_type);
  if (order_mode < 0 || order_mode > 4 || (order_mode == 4 && !perm_in))
    return set_err(GSPX_ERR_INVALID, "order_mode: 0 none, 1 auto, 2 morton, 3 hilbert, 4 given permutation");
  if (N < 0 || nnz < 0 || N >= ((int64_t)1 << 30) || nnz >= ((int64_t)1 << 31) - 8 * N - 64)
    return set_err(GSPX_ERR_INVALID, "graph too large for int32 indexing");
  if (!indptr || (nnz > 0 && (!indices || !data))) return set_err(GSPX_ERR_INVALID, "null CSR arrays");
  if (indptr[0] != 0 || indptr[N] != nnz)
    return set_err(GSPX_ERR_INVALID, "indptr[0] must be 0 and indptr[N] must equal nnz");
  HIPCHK(hipSetDevice(ctx->device));
  gspx_graph* g = new gspx_graph();
  g->ctx = ctx;
  g->N = N;
  g->dtype = compute_dtype;
  g->from_w = true;
  g->lap_type = lap_type;
  int rc;
#define GSPX_SETUP(TIN)                                                                                       \
  (compute_dtype == GSPX_F32                                                                                  \
       ? graph_setup_t<TIN, float>(g, nnz, indptr, indices, data, coords, d, order_mode, perm_in, report)   \
       : graph_setup_t<TIN, double>(g, nnz, indptr, indices, data, coords, d, order_mode, perm_in, report))
  if (data_dtype == GSPX_F32) rc = GSPX_SETUP(float);
  else if (data_dtype == GSPX_F64) rc = GSPX_SETUP(double);
  else rc = GSPX_SETUP(long long);
#undef GSPX_SETUP
  if (rc != GSPX_OK || report[10] != 0) {
    delete g;
    return rc;
  }
  *out = g;
  return GSPX_OK;
}

// The same set-up for a W that a device builder left on the device (gspx_knn_build / gspx_radius_build /
// gspx_sbm_build): no download, no upload - the generator classes hand the builder's handle over and the host copy of
// W is made only when somebody reads G.W.
extern "C" int gspx_graph_setup_from_knn(gspx_knn* h, int lap_type, int compute_dtype, const double* coords, int d,
                                         int order_mode, const int32_t* perm_in, int64_t report[12], gspx_graph** out) {
  if (!h || !out || !report) return set_err(GSPX_ERR_INVALID, "null handle, report or output");
  *out = nullptr;
  for (int i = 0; i < 12; ++i) report[i] = 0;
  if (compute_dtype != GSPX_F32 && compute_dtype != GSPX_F64)
    return set_err(GSPX_ERR_INVALID, "compute_dtype must be GSPX_F32 or GSPX_F64");
  if (lap_type != GSPX_LAP_COMBINATORIAL && lap_type != GSPX_LAP_NORMALIZED)
    return set_err(GSPX_ERR_INVALID, "Unknown Laplacian type %d", lap_type);
  if (order_mode < 0 || order_mode > 4 || (order_mode == 4 && !perm_in))
    return set_err(GSPX_ERR_INVALID, "order_mode: 0 none, 1 auto, 2 morton, 3 hilbert, 4 given permutation");
  if (coords && d != 0 && (d < 1 || d > 64)) return set_err(GSPX_ERR_INVALID, "coords: 1 to 64 dimensions");
  const int64_t N = h->N, nnz = h->nnz;
  if (N >= ((int64_t)1 << 30) || nnz >= ((int64_t)1 << 31) - 8 * N - 64)
    return set_err(GSPX_ERR_INVALID, "graph too large for int32 indexing");
  if (!h->rowptr.p || (nnz > 0 && (!h->col.p || !h->val.p))) return set_err(GSPX_ERR_INVALID, "the handle holds no W");
  gspx_ctx* ctx = h->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  replay_reset(ctx);
  gspx_graph* g = new gspx_graph();
  g->ctx = ctx;
  g->N = N;
  g->dtype = compute_dtype;
  g->from_w = true;
  g->lap_type = lap_type;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = compute_dtype == GSPX_F32
                     ? graph_setup_core<double, float>(g, nnz, h->rowptr.as<int>(), h->col.as<int>(), h->val.as<double>(),
                                                       coords, d, order_mode, perm_in, report, t0)
                     : graph_setup_core<double, double>(g, nnz, h->rowptr.as<int>(), h->col.as<int>(), h->val.as<double>(),
                                                        coords, d, order_mode, perm_in, report, t0);
  if (rc != GSPX_OK || report[10] != 0) {
    delete g;
    return rc;
  }
  *out = g;
  return GSPX_OK;
}

extern "C" int gspx_graph_download_perm(gspx_graph* g, int32_t* perm) {
  if (!g || (!perm && g->N > 0)) return set_err(GSPX_ERR_INVALID, "null argument");
  if (!g->has_perm) return set_err(GSPX_ERR_INVALID, "the graph has no internal vertex order");
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpy(perm, g->perm.p, (size_t)g->N * sizeof(int), hipMemcpyDeviceToHost));
  return GSPX_OK;
}
