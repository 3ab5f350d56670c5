// gspx_knn.hip.h - k-nearest-neighbour graph construction on the device.
// Included at the end of gspx.hip.  gfx950 only.
//
// Replaces, for NNtype='knn' (1 <= d <= 3 by the grid search below; 4 <= d <= 64 by the tiled brute force on the
// matrix cores of gspx_knn_bf.hip.h; the weights / symmetrisation / CSR stages are shared):
//   pygsp/graphs/nngraphs/nngraph.py:213-216  kdt = spatial.KDTree(Xout); D, NN = kdt.query(Xout, k + 1)
//   nngraph.py:218-226                        sigma = mean(D[:, 1:]);  w = exp(-D^2 / sigma)
//   nngraph.py:289-297                        W = csc((w, (i, j)));  W = (W + W.T) / 2
// Method: uniform grid (about 2.5 points per cell), counting sort of the points by cell, one thread
// per query walking the Chebyshev rings of cells around its own until the k-th best distance is
// below what any unvisited ring can offer.  Distances are formed exactly like the KD-tree's
// (sum of squared differences in dimension order, no fused multiply-add, one correctly rounded
// sqrt), so neighbour lists and distances equal scipy's bit for bit (ties, i.e. exactly equal
// distances, are ordered by vertex index).
#pragma once

namespace gspx {

struct KnnGrid {
  double lo[3];
  double inv_h[3];
  double h_min;
  int n[3];
  int d;
  int metric;  // 0 euclidean, 1 manhattan, 2 max_dist (the reference's dist_type, nngraph.py:139-145)
};

__device__ __forceinline__ int knn_cell_of(const KnnGrid& g, const double* __restrict__ p, int c[3]) {
  int id = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    c[j] = 0;
    if (j < g.d) {
      int v = (int)((p[j] - g.lo[j]) * g.inv_h[j]);
      v = v < 0 ? 0 : (v >= g.n[j] ? g.n[j] - 1 : v);
      c[j] = v;
    }
  }
  id = (c[2] * g.n[1] + c[1]) * g.n[0] + c[0];
  return id;
}

__global__ void k_knn_cell_count(const double* __restrict__ x, int N, KnnGrid g, int* __restrict__ cell,
                                 int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[3];
  const int id = knn_cell_of(g, x + (size_t)i * g.d, c);
  cell[i] = id;
  atomicAdd(&count[id], 1);
}
__global__ void k_knn_scatter(const int* __restrict__ cell, int N, const int* __restrict__ start,
                              int* __restrict__ cursor, int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int id = cell[i];
  order[start[id] + atomicAdd(&cursor[id], 1)] = i;
}
// the scatter's order inside a cell depends on atomics: sort every cell's points by index, and
// gather the coordinates in that order (sorted[s] = x[order[s]])
__global__ void k_knn_cell_sort(const int* __restrict__ start, int ncells, int* __restrict__ order,
                                const double* __restrict__ x, int d, double* __restrict__ sorted) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const int lo = start[c], hi = start[c + 1];
  for (int a = lo + 1; a < hi; ++a) {
    const int v = order[a];
    int b = a - 1;
    while (b >= lo && order[b] > v) {
      order[b + 1] = order[b];
      --b;
    }
    order[b + 1] = v;
  }
  for (int a = lo; a < hi; ++a)
    for (int j = 0; j < d; ++j) sorted[(size_t)a * d + j] = x[(size_t)order[a] * d + j];
}

// KD-tree arithmetic (scipy ckdtree, distance_base.h: sqeuclidean_distance_double): four running sums over the
// dimensions taken four at a time - acc[j] += (u[i+j] - v[i+j])^2 -, then s = ((acc0 + acc1) + acc2) + acc3, then
// the remaining (at most three) dimensions added to s one by one; in 1-3 dimensions that is the plain sum in
// dimension order.  Every operation is rounded on its own - hipcc contracts a*b+c into an fma by default, which
// changes the last bit, so contraction is switched off here.  (Checked against scipy 1.15 in 4 to 64
// dimensions: tests/test_gpu_5_knn.py::test_highdim_oracle.)
__device__ __forceinline__ double knn_sqdist(const double* q, const double* __restrict__ p, int d) {
#pragma clang fp contract(off)
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int j = 0;
  for (; j + 4 <= d; j += 4) {
    const double d0 = q[j] - p[j], d1 = q[j + 1] - p[j + 1], d2 = q[j + 2] - p[j + 2], d3 = q[j + 3] - p[j + 3];
    const double s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
    a0 = a0 + s0;
    a1 = a1 + s1;
    a2 = a2 + s2;
    a3 = a3 + s3;
  }
  double d2 = ((a0 + a1) + a2) + a3;
  for (; j < d; ++j) {
    const double df = q[j] - p[j];
    const double sq = df * df;
    d2 = d2 + sq;
  }
  return d2;
}
// the comparison key of a candidate: squared euclidean distance, or the manhattan / max distance itself
// (ckdtree's MinkowskiDistP1 / Pinf: a sum of |differences| in dimension order / their maximum - exact)
__device__ __forceinline__ double knn_key(const double* q, const double* __restrict__ p, int d, int metric) {
  if (metric == 0) return knn_sqdist(q, p, d);
  double s = 0;
  for (int j = 0; j < d; ++j) {
    const double a = fabs(q[j] - p[j]);
    s = metric == 1 ? s + a : fmax(s, a);
  }
  return s;
}
// correctly rounded square root: the hardware-assisted sqrt is within an ulp; one exact residual
// (fma) and a correction decide the last bit
__device__ __forceinline__ double knn_sqrt(double x) {
  if (!(x > 0.0)) return 0.0;
  const double s = sqrt(x);
  const double r = fma(-s, s, x);  // exact x - s*s
  return s + r / (2.0 * s);
}

// one thread per query (in cell order: a wave's queries are neighbours in space).  best[] is kept
// sorted by (distance, index) through a chain of conditional swaps - no dynamic register indexing.
template <int KMAX>
__global__ __launch_bounds__(128) void k_knn_query(const double* __restrict__ sorted,
                                                   const int* __restrict__ order,
                                                   const int* __restrict__ start, int N, KnnGrid g,
                                                   int k, int* __restrict__ nn,
                                                   double* __restrict__ dist) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int self = order[s];
  double q[3] = {0, 0, 0};
  for (int j = 0; j < g.d; ++j) q[j] = sorted[(size_t)s * g.d + j];
  int c[3];
  knn_cell_of(g, q, c);
  // distance from the query to the nearest wall of its own cell (>= 0 up to rounding)
  double m = 1e300;
  for (int j = 0; j < g.d; ++j) {
    const double h = 1.0 / g.inv_h[j];
    const double a = q[j] - (g.lo[j] + c[j] * h), b = (g.lo[j] + (c[j] + 1) * h) - q[j];
    m = fmin(m, fmin(a, b));
  }
  m = fmax(m, 0.0);
  double bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) {
    bd[t] = 1e300;
    bi[t] = 0x7fffffff;
  }
  const int rmax = max(g.n[0], max(g.n[1], g.n[2]));
  const int r2 = g.d >= 2 ? 1 : 0, r3 = g.d >= 3 ? 1 : 0;
  for (int r = 0; r <= rmax; ++r) {
    for (int dz = -r * r3; dz <= r * r3; ++dz) {
      const int cz = c[2] + dz;
      if (cz < 0 || cz >= g.n[2]) continue;
      for (int dy = -r * r2; dy <= r * r2; ++dy) {
        const int cy = c[1] + dy;
        if (cy < 0 || cy >= g.n[1]) continue;
        const bool shell = (abs(dz) == r && r3) || (abs(dy) == r && r2);
        // on the shell every dx counts; inside it only the two end cells dx = -r, +r
        const int step = (shell || r == 0) ? 1 : 2 * r;
        for (int dx = -r; dx <= r; dx += step) {
          const int cx = c[0] + dx;
          if (cx < 0 || cx >= g.n[0]) continue;
          const int cid = (cz * g.n[1] + cy) * g.n[0] + cx;
          for (int a = start[cid]; a < start[cid + 1]; ++a) {
            const int idx = order[a];
            if (idx == self) continue;
            const double d2 = knn_key(q, sorted + (size_t)a * g.d, g.d, g.metric);
            if (d2 < bd[KMAX - 1] || (d2 == bd[KMAX - 1] && idx < bi[KMAX - 1])) {
              double cd = d2;
              int ci = idx;
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
        }
      }
    }
    // every point not yet visited lies beyond r whole cells plus the way out of the own cell
    double kth = 1e300;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
      if (t == k - 1) kth = bd[t];
    const double lb = (r * g.h_min + m) * (1.0 - 1e-12);
    if (kth < 1e300 && kth <= (g.metric == 0 ? lb * lb : lb)) break;  // every p-norm is >= the max norm
  }
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t < k) {
      nn[(size_t)self * k + t] = bi[t];
      dist[(size_t)self * k + t] = g.metric == 0 ? knn_sqrt(bd[t]) : bd[t];
    }
}

// sum of all distances (double, fixed order): partial sums per block, then one block
__global__ __launch_bounds__(256) void k_knn_sum_partial(const double* __restrict__ v, size_t n,
                                                         double* __restrict__ partial) {
  __shared__ double ws[4];
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// w = exp(-d^2 / sigma) (nngraph.py:224-226: the distance is squared again after the sqrt); an edge
// i -> j is mutual when i is among j's neighbours; non-mutual edges add one entry to row j
// Symmetrisation of the directed k-NN matrix (utils.symmetrize, utils.py:247-275; the weights of a
// mutual pair are equal):  0 'average'  (W + W^T) / 2 - a mutual pair keeps w, a one-sided edge leaves
// w / 2 in both rows;  1 'maximum' / 'fill' - w in both rows either way;  2 'tril' / 3 'triu' - only
// the entries below / above the diagonal count: the pair {i, j} exists iff its larger (smaller) end
// chose the other.  Per directed edge i -> j: `self` = row i keeps (i, j), `other` = row j gets (j, i).
__device__ __forceinline__ void knn_sym_rule(int sym, int i, int j, bool mutual, bool& self, bool& other,
                                             double& scale) {
  scale = 1.0;
  if (sym <= 1) {
    self = true;
    other = !mutual;
    if (sym == 0 && !mutual) scale = 0.5;
  } else {
    self = other = (sym == 2) ? (j < i) : (j > i);
  }
}
// w = exp(-d^2 / sigma) (nngraph.py:224-226: the distance is squared again after the sqrt); an edge
// i -> j is mutual when i is among j's neighbours; row lengths are counted here
__global__ void k_knn_weights(const int* __restrict__ nn, const double* __restrict__ dist, int N, int k,
                              double sigma, int sym, double* __restrict__ w,
                              unsigned char* __restrict__ mutual, int* __restrict__ len) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)N * k) return;
  const int i = (int)(e / k);
  const int j = nn[e];
  const double d = dist[e];
  w[e] = exp(-(d * d) / sigma);
  bool mu = false;
  for (int t = 0; t < k; ++t) mu |= nn[(size_t)j * k + t] == i;
  mutual[e] = mu ? 1 : 0;
  bool self, other;
  double scale;
  knn_sym_rule(sym, i, j, mu, self, other, scale);
  if (self) atomicAdd(&len[i], 1);
  if (other) atomicAdd(&len[j], 1);
}
// The same result out of place, without dependent memory chains: every entry's final position is the number of
// smaller columns in its row (columns are distinct within a row), counted with independent, cache-resident loads.
// (The in-place insertion sort above shifts through global memory - a dependent load / store pair per shift: it
// was 15 % of a high-dimensional build.)
__global__ void k_knn_row_rank(const int* __restrict__ rowptr, int N, const int* __restrict__ col,
                               const double* __restrict__ val, int* __restrict__ col2, double* __restrict__ val2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int lo = rowptr[i], hi = rowptr[i + 1];
  if (hi - lo > 64) return;  // long rows (hubs - common in high dimensions): a whole wave each, below
  for (int a = lo; a < hi; ++a) {
    const int c = col[a];
    int rank = 0;
    for (int b = lo; b < hi; ++b) rank += col[b] < c;
    col2[lo + rank] = c;
    val2[lo + rank] = val[a];
  }
}
// rows longer than 64 entries: one wave per row, every lane ranks its share of the entries
__global__ __launch_bounds__(256) void k_knn_row_rank_long(const int* __restrict__ rowptr, int N,
                                                           const int* __restrict__ col, const double* __restrict__ val,
                                                           int* __restrict__ col2, double* __restrict__ val2) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const int lo = rowptr[i], hi = rowptr[i + 1];
  if (hi - lo <= 64) return;
  for (int a = lo + lane; a < hi; a += 64) {
    const int c = col[a];
    int rank = 0;
    for (int b = lo; b < hi; ++b) rank += col[b] < c;
    col2[lo + rank] = c;
    val2[lo + rank] = val[a];
  }
}
__global__ void k_knn_fill(const int* __restrict__ nn, const double* __restrict__ w,
                           const unsigned char* __restrict__ mutual, int N, int k, int sym,
                           const int* __restrict__ rowptr, int* __restrict__ cursor,
                           int* __restrict__ col, double* __restrict__ val) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)N * k) return;
  const int i = (int)(e / k);
  const int j = nn[e];
  bool self, other;
  double scale;
  knn_sym_rule(sym, i, j, mutual[e] != 0, self, other, scale);
  const double v = w[e] * scale;
  if (self) {
    const int o = rowptr[i] + atomicAdd(&cursor[i], 1);
    col[o] = j;
    val[o] = v;
  }
  if (other) {
    const int o = rowptr[j] + atomicAdd(&cursor[j], 1);
    col[o] = i;
    val[o] = v;
  }
}
__global__ void k_knn_row_sort(const int* __restrict__ rowptr, int N, int* __restrict__ col,
                               double* __restrict__ val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int lo = rowptr[i], hi = rowptr[i + 1];
  for (int a = lo + 1; a < hi; ++a) {
    const int c = col[a];
    const double v = val[a];
    int b = a - 1;
    while (b >= lo && col[b] > c) {
      col[b + 1] = col[b];
      val[b + 1] = val[b];
      --b;
    }
    col[b + 1] = c;
    val[b + 1] = v;
  }
}

// space-filling-curve key of every point (internal vertex order = argsort of the keys):
// curve 0 = Morton (bit interleave, 31 bits per axis in 2-D, 21 in 3-D), 1 = Hilbert (2-D, 16 bits
// per axis; the classic xy -> d walk)
__global__ void k_curve_keys(const double* __restrict__ x, int N, int d, double lo0, double lo1, double lo2,
                             double inv0, double inv1, double inv2, int curve,
                             unsigned long long* __restrict__ key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double lo[3] = {lo0, lo1, lo2}, inv[3] = {inv0, inv1, inv2};
  if (curve == 1) {
    const unsigned long long top = (1ull << 16) - 1;
    long long q[2];
    for (int j = 0; j < 2; ++j) {
      long long v = (long long)((x[(size_t)i * d + j] - lo[j]) * inv[j] * 65536.0);
      q[j] = v < 0 ? 0 : (v > (long long)top ? (long long)top : v);
    }
    long long px = q[0], py = q[1];
    unsigned long long dd = 0;
    for (long long s = 1ll << 15; s > 0; s >>= 1) {
      const long long rx = (px & s) ? 1 : 0, ry = (py & s) ? 1 : 0;
      dd += (unsigned long long)(s * s) * (unsigned long long)((3 * rx) ^ ry);
      if (ry == 0) {
        if (rx == 1) {
          px = s - 1 - px;
          py = s - 1 - py;
        }
        const long long t = px;
        px = py;
        py = t;
      }
    }
    key[i] = dd;
    return;
  }
  const int dm = d >= 3 ? 3 : 2;
  const int bits = dm == 3 ? 21 : 31;
  const double scale = (double)((1ull << bits) - 1);
  unsigned long long code = 0;
  unsigned long long q[3] = {0, 0, 0};
  for (int j = 0; j < dm; ++j) {
    double v = (x[(size_t)i * d + j] - lo[j]) * inv[j] * scale;
    v = v < 0 ? 0 : (v > scale ? scale : v);
    q[j] = (unsigned long long)v;
  }
  for (int b = 0; b < bits; ++b)
    for (int j = 0; j < dm; ++j) code |= ((q[j] >> b) & 1ull) << (b * dm + j);
  key[i] = code;
}

}  // namespace gspx

// keys of N host points into a device buffer (queued on the context's stream; returns after the launch)
static int curve_keys_dev(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve, unsigned long long* keys_dev) {
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, inv[3] = {1, 1, 1};
  const int dm = std::min(d, 3);
  for (int j = 0; j < dm; ++j) lo[j] = hi[j] = coords[j];
  for (int64_t i = 0; i < N; ++i)
    for (int j = 0; j < dm; ++j) {
      const double v = coords[i * d + j];
      if (!std::isfinite(v)) return set_err(GSPX_ERR_INVALID, "non-finite coordinate");
      lo[j] = std::min(lo[j], v);
      hi[j] = std::max(hi[j], v);
    }
  for (int j = 0; j < dm; ++j) inv[j] = hi[j] > lo[j] ? 1.0 / (hi[j] - lo[j]) : 1.0;
  DevMem x;
  CHK(x.alloc((size_t)N * d * sizeof(double)));
  HIPCHK(hipMemcpyAsync(x.p, coords, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_curve_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, x.as<double>(),
                     (int)N, d, lo[0], lo[1], lo[2], inv[0], inv[1], inv[2], curve, keys_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));  // x goes out of scope
  return GSPX_OK;
}

extern "C" int gspx_curve_keys(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve,
                               uint64_t* keys) {
  if (!ctx || !coords || !keys) return set_err(GSPX_ERR_INVALID, "null argument");
  if (N < 1 || N >= ((int64_t)1 << 31) || d < 2) return set_err(GSPX_ERR_INVALID, "gspx_curve_keys: bad N or d");
  if (curve != 0 && curve != 1) return set_err(GSPX_ERR_INVALID, "curve: 0 Morton, 1 Hilbert");
  HIPCHK(hipSetDevice(ctx->device));
  DevMem k;
  CHK(k.alloc((size_t)N * sizeof(uint64_t)));
  CHK(curve_keys_dev(ctx, N, d, coords, curve, (unsigned long long*)k.p));
  HIPCHK(hipMemcpy(keys, k.p, (size_t)N * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return GSPX_OK;
}

struct gspx_knn {
  gspx_ctx* ctx = nullptr;
  int64_t N = 0;
  int d = 0, k = 0;
  double sigma = 0.0;
  int64_t nnz = 0;
  double build_ms = 0.0;
  double search_stats[4] = {0, 0, 0, 0};  // brute-force search (d > 3): sample, capacity, mean candidates, exact scans
  DevMem nn, dist, rowptr, col, val;
};

#include "gspx_knn_bf.hip.h"


template <int KMAX>
static void launch_knn_query(const double* sorted, const int* order, const int* start, int N,
                             const KnnGrid& g, int k, int* nn, double* dist, hipStream_t st) {
  hipLaunchKernelGGL((k_knn_query<KMAX>), dim3((N + 127) / 128), dim3(128), 0, st, sorted, order, start, N,
                     g, k, nn, dist);
}

extern "C" int gspx_knn_build(gspx_ctx* ctx, int64_t N, int d, const double* coords, int k,
                              double sigma, int metric, int symmetrize, gspx_knn** out) {
  if (!ctx || !out) return set_err(GSPX_ERR_INVALID, "null ctx or output");
  *out = nullptr;
  if (metric < 0 || metric > 2) return set_err(GSPX_ERR_INVALID, "metric: 0 euclidean, 1 manhattan, 2 max_dist");
  if (symmetrize < 0 || symmetrize > 3)
    return set_err(GSPX_ERR_INVALID, "symmetrize: 0 average, 1 maximum / fill, 2 tril, 3 triu");
  if (N < 2 || N >= ((int64_t)1 << 31) / 64) return set_err(GSPX_ERR_INVALID, "gspx_knn_build: bad N");
  if (d < 1 || d > 64)
    return set_err(GSPX_ERR_INVALID, "gspx_knn_build: the device k-NN search covers 1 to 64 dimensions (got %d)", d);
  if (k < 1 || k > 64) return set_err(GSPX_ERR_INVALID, "gspx_knn_build: 1 <= k <= 64 (got %d)", k);
  if (k >= N)  // nngraph.py:123-127
    return set_err(GSPX_ERR_INVALID, "The number of neighbors (k=%d) must be smaller than the number of nodes (%lld).",
                   k, (long long)N);
  if (!coords) return set_err(GSPX_ERR_INVALID, "null coordinates");
  if (!(sigma >= 0) || !std::isfinite(sigma)) return set_err(GSPX_ERR_INVALID, "sigma must be >= 0 (0: mean distance)");
  KnnGrid g{};
  const bool grid_search = d <= 3;  // beyond three dimensions: tiled brute force on the matrix cores (gspx_knn_bf.hip.h)
  const int dg = grid_search ? d : 0;
  g.d = dg;
  g.metric = metric;
  double hi[3] = {0, 0, 0};
  for (int j = 0; j < 3; ++j) {
    g.lo[j] = 0;
    g.inv_h[j] = 1;
    g.n[j] = 1;
  }
  for (int j = 0; j < dg; ++j) {
    g.lo[j] = coords[j];
    hi[j] = coords[j];
  }
  for (int64_t i = 0; i < N; ++i)
    for (int j = 0; j < d; ++j) {
      const double v = coords[i * d + j];
      if (!std::isfinite(v)) return set_err(GSPX_ERR_INVALID, "non-finite coordinate");
      if (j < dg) {
        g.lo[j] = std::min(g.lo[j], v);
        hi[j] = std::max(hi[j], v);
      }
    }
  // about 2.5 points per cell over the bounding box; degenerate extents get one cell
  const int cap = d == 1 ? (1 << 21) : (d == 2 ? 2048 : 128);
  const int per_dim = std::max(1, std::min(cap, (int)std::floor(std::pow((double)N / 2.5, 1.0 / std::max(dg, 1)))));
  g.h_min = 1e300;
  int64_t ncells = 1;
  for (int j = 0; j < dg; ++j) {
    const double ext = hi[j] - g.lo[j];
    g.n[j] = ext > 0 ? per_dim : 1;
    const double h = ext > 0 ? ext / g.n[j] : 1.0;
    g.inv_h[j] = 1.0 / h;
    if (ext > 0) g.h_min = std::min(g.h_min, h);
    ncells *= g.n[j];
  }
  if (g.h_min == 1e300) g.h_min = 1.0;  // all points coincide
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto t0 = std::chrono::steady_clock::now();
  gspx_knn* h = new gspx_knn();
  h->ctx = ctx;
  h->N = N;
  h->d = d;
  h->k = k;
  auto fail = [&](int rc) {
    delete h;
    return rc;
  };
#define KCHK(x)                      \
  do {                               \
    int rc__ = (x);                  \
    if (rc__ != GSPX_OK) return fail(rc__); \
  } while (0)
#define KHIP(x)                                                                            \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) return fail(set_err(GSPX_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__))); \
  } while (0)
  const int n = (int)N;
  const size_t nk = (size_t)N * k;
  DevMem x, sorted, cell, count, start, cursor, order, w, mutual, len, partial;
  KCHK(x.alloc((size_t)N * d * sizeof(double)));
  KCHK(sorted.alloc((size_t)N * d * sizeof(double)));
  KCHK(cell.alloc((size_t)N * sizeof(int)));
  KCHK(count.alloc(((size_t)ncells + 1) * sizeof(int)));
  KCHK(start.alloc(((size_t)ncells + 1) * sizeof(int)));
  KCHK(cursor.alloc(((size_t)std::max<int64_t>(ncells, N) + 1) * sizeof(int)));
  KCHK(order.alloc((size_t)N * sizeof(int)));
  KCHK(h->nn.alloc(nk * sizeof(int)));
  KCHK(h->dist.alloc(nk * sizeof(double)));
  KHIP(hipMemcpyAsync(x.p, coords, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, st));
  KHIP(hipMemsetAsync(count.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
  KHIP(hipMemsetAsync(cursor.p, 0, ((size_t)std::max<int64_t>(ncells, N) + 1) * sizeof(int), st));
  const int nbN = (n + 255) / 256;
  if (!grid_search) {
    KHIP(hipStreamSynchronize(st));
    KCHK(knn_bruteforce(ctx, x.as<double>(), n, d, k, metric, h->nn.as<int>(), h->dist.as<double>(), h->search_stats));
  } else {
  hipLaunchKernelGGL(k_knn_cell_count, dim3(nbN), dim3(256), 0, st, x.as<double>(), n, g, cell.as<int>(),
                     count.as<int>());
  KCHK(scan_exclusive(ctx, count.as<int>(), start.as<int>(), (int)ncells + 1));
  hipLaunchKernelGGL(k_knn_scatter, dim3(nbN), dim3(256), 0, st, cell.as<int>(), n, start.as<int>(),
                     cursor.as<int>(), order.as<int>());
  hipLaunchKernelGGL(k_knn_cell_sort, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st,
                     start.as<int>(), (int)ncells, order.as<int>(), x.as<double>(), d, sorted.as<double>());
  if (k <= 8)
    launch_knn_query<8>(sorted.as<double>(), order.as<int>(), start.as<int>(), n, g, k, h->nn.as<int>(),
                        h->dist.as<double>(), st);
  else if (k <= 16)
    launch_knn_query<16>(sorted.as<double>(), order.as<int>(), start.as<int>(), n, g, k, h->nn.as<int>(),
                         h->dist.as<double>(), st);
  else if (k <= 32)
    launch_knn_query<32>(sorted.as<double>(), order.as<int>(), start.as<int>(), n, g, k, h->nn.as<int>(),
                         h->dist.as<double>(), st);
  else
    launch_knn_query<64>(sorted.as<double>(), order.as<int>(), start.as<int>(), n, g, k, h->nn.as<int>(),
                         h->dist.as<double>(), st);
  }
  KHIP(hipGetLastError());
  // sigma = mean neighbour distance (nngraph.py:218-219) unless given
  if (sigma == 0.0) {
    const int nb = 1024;
    KCHK(partial.alloc((size_t)nb * sizeof(double)));
    hipLaunchKernelGGL(k_knn_sum_partial, dim3(nb), dim3(256), 0, st, h->dist.as<double>(), nk,
                       partial.as<double>());
    std::vector<double> hp(nb);
    KHIP(hipMemcpyAsync(hp.data(), partial.p, nb * sizeof(double), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    double s = 0;
    for (double v : hp) s += v;
    sigma = s / (double)nk;
    if (!(sigma > 0)) return fail(set_err(GSPX_ERR_INVALID, "gspx_knn_build: all neighbour distances are zero"));
  }
  h->sigma = sigma;
  KCHK(w.alloc(nk * sizeof(double)));
  KCHK(mutual.alloc(nk));
  KCHK(len.alloc(((size_t)N + 1) * sizeof(int)));
  KCHK(h->rowptr.alloc(((size_t)N + 1) * sizeof(int)));
  const unsigned nbE = (unsigned)((nk + 255) / 256);
  KHIP(hipMemsetAsync(len.p, 0, ((size_t)N + 1) * sizeof(int), st));
  hipLaunchKernelGGL(k_knn_weights, dim3(nbE), dim3(256), 0, st, h->nn.as<int>(), h->dist.as<double>(), n, k,
                     sigma, symmetrize, w.as<double>(), mutual.as<unsigned char>(), len.as<int>());
  KCHK(scan_exclusive(ctx, len.as<int>(), h->rowptr.as<int>(), n + 1));
  int nnz = 0;
  KHIP(hipMemcpyAsync(&nnz, h->rowptr.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
  KHIP(hipStreamSynchronize(st));
  h->nnz = nnz;
  KCHK(h->col.alloc((size_t)std::max(nnz, 1) * sizeof(int)));
  KCHK(h->val.alloc((size_t)std::max(nnz, 1) * sizeof(double)));
  KHIP(hipMemsetAsync(cursor.p, 0, ((size_t)N + 1) * sizeof(int), st));
  hipLaunchKernelGGL(k_knn_fill, dim3(nbE), dim3(256), 0, st, h->nn.as<int>(), w.as<double>(),
                     mutual.as<unsigned char>(), n, k, symmetrize, h->rowptr.as<int>(), cursor.as<int>(), h->col.as<int>(),
                     h->val.as<double>());
  {
    DevMem col2, val2;
    KCHK(col2.alloc((size_t)std::max(nnz, 1) * sizeof(int)));
    KCHK(val2.alloc((size_t)std::max(nnz, 1) * sizeof(double)));
    hipLaunchKernelGGL(k_knn_row_rank, dim3(nbN), dim3(256), 0, st, h->rowptr.as<int>(), n, h->col.as<int>(),
                       h->val.as<double>(), col2.as<int>(), val2.as<double>());
    hipLaunchKernelGGL(k_knn_row_rank_long, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, h->rowptr.as<int>(), n,
                       h->col.as<int>(), h->val.as<double>(), col2.as<int>(), val2.as<double>());
    KHIP(hipGetLastError());
    KHIP(hipStreamSynchronize(st));
    std::swap(h->col.p, col2.p);      // the sorted arrays become the result; the unsorted ones go with col2 / val2
    std::swap(h->col.bytes, col2.bytes);
    std::swap(h->val.p, val2.p);
    std::swap(h->val.bytes, val2.bytes);
  }
#undef KCHK
#undef KHIP
  h->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  return GSPX_OK;
}

extern "C" int gspx_knn_destroy(gspx_knn* h) {
  if (!h) return GSPX_OK;
  (void)hipSetDevice(h->ctx->device);
  delete h;
  return GSPX_OK;
}

extern "C" int gspx_knn_info(gspx_knn* h, int64_t* nnz, double* sigma, double* build_ms) {
  if (!h) return set_err(GSPX_ERR_INVALID, "null handle");
  if (nnz) *nnz = h->nnz;
  if (sigma) *sigma = h->sigma;
  if (build_ms) *build_ms = h->build_ms;
  return GSPX_OK;
}

extern "C" int gspx_knn_search_stats(gspx_knn* h, double out[4]) {
  if (!h || !out) return set_err(GSPX_ERR_INVALID, "null argument");
  for (int i = 0; i < 4; ++i) out[i] = h->search_stats[i];
  return GSPX_OK;
}

extern "C" int gspx_knn_download_w(gspx_knn* h, int32_t* indptr, int32_t* indices, double* data) {
  if (!h || !indptr) return set_err(GSPX_ERR_INVALID, "null argument");
  if (h->nnz > 0 && (!indices || !data)) return set_err(GSPX_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->ctx->device));
  HIPCHK(hipMemcpy(indptr, h->rowptr.p, ((size_t)h->N + 1) * sizeof(int), hipMemcpyDeviceToHost));
  if (h->nnz > 0) {
    HIPCHK(hipMemcpy(indices, h->col.p, (size_t)h->nnz * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(data, h->val.p, (size_t)h->nnz * sizeof(double), hipMemcpyDeviceToHost));
  }
  return GSPX_OK;
}

extern "C" int gspx_knn_download_neighbors(gspx_knn* h, int32_t* nn, double* dist) {
  if (!h) return set_err(GSPX_ERR_INVALID, "null handle");
  if (h->k == 0) return set_err(GSPX_ERR_INVALID, "this handle was not built by gspx_knn_build");
  HIPCHK(hipSetDevice(h->ctx->device));
  const size_t nk = (size_t)h->N * h->k;
  if (nn) HIPCHK(hipMemcpy(nn, h->nn.p, nk * sizeof(int), hipMemcpyDeviceToHost));
  if (dist) HIPCHK(hipMemcpy(dist, h->dist.p, nk * sizeof(double), hipMemcpyDeviceToHost));
  return GSPX_OK;
}

// ------------------------------------------------------------------------------------------------
// Stochastic block model / Erdos-Renyi sampler on the device (second half).
// The reference visits all N^2 ordered pairs in a Python loop and keeps pair (r > c) with probability
// M[z_r][z_c] (pygsp/graphs/stochasticblockmodel.py:125-144; erdosrenyi.py is the one-block case).
// Same distribution in O(edges): the candidates of every block pair (a, b <= a) - a triangle of
// n_a (n_a - 1) / 2 pairs for a == b, a rectangle of n_a n_b otherwise - are cut into chunks; one
// thread walks a chunk by geometric skips (the gap to the next kept pair of independent Bernoulli(p)
// trials is Geometric(p)), so every pair is kept independently with probability p, chunks are
// independent, and no duplicate can arise.  Counter-based random numbers (a hash of seed, chunk and
// draw number): pass 1 counts, pass 2 replays the same stream and emits.  The stream differs from
// numpy's, so graphs are equal in distribution to the reference's, not bit-equal.
// ------------------------------------------------------------------------------------------------
namespace gspx {

struct SbmSeg {       // one block pair
  long long total;    // candidate pairs
  long long chunk;    // candidates per chunk
  long long first;    // global index of its first chunk
  double log1mp;      // log(1 - p)  (-inf for p == 1)
  int lo_a, n_a, lo_b, n_b;  // member ranges in `order`
  int kind;           // candidate index -> (r, c): 0 rectangle n_a x n_b, 1 triangle r > c, 2 triangle r >= c
                      // (self-loops allowed), 3 square n_a x n_a without its diagonal (directed, no self-loops)
};
constexpr int GSPX_SBM_DIRECTED = 1, GSPX_SBM_SELF_LOOPS = 2;

__device__ __forceinline__ unsigned long long sbm_hash(unsigned long long x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// u in (0, 1]
__device__ __forceinline__ double sbm_uniform(unsigned long long seed, unsigned long long chunk, unsigned draw) {
  const unsigned long long h = sbm_hash(sbm_hash(seed ^ (chunk * 0xD1B54A32D192ED03ull)) + draw);
  return (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0);
}

// PASS 0: count per chunk; PASS 1: emit (er[], ec[]) at off[chunk] and count degrees
template <int PASS>
__global__ void k_sbm_chunks(const SbmSeg* __restrict__ seg, int nseg, long long nchunks,
                             unsigned long long seed, const int* __restrict__ order,
                             int* __restrict__ cnt, const int* __restrict__ off, int* __restrict__ er,
                             int* __restrict__ ec, int* __restrict__ deg, int directed) {
  const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= nchunks) return;
  int lo = 0, hi = nseg - 1;  // last segment whose first chunk is <= ch
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid].first <= ch) lo = mid; else hi = mid - 1;
  }
  const SbmSeg s = seg[lo];
  const long long base = (ch - s.first) * s.chunk;
  const long long len = min(s.chunk, s.total - base);
  long long pos = 0;
  int n = 0;
  const int o = PASS ? off[ch] : 0;
  for (unsigned draw = 0;; ++draw) {
    const double u = sbm_uniform(seed, (unsigned long long)ch, draw);
    const double g = floor(log(u) / s.log1mp);  // failures before the next success
    if (!(g < (double)(len - pos))) break;
    pos += (long long)g;
    if (PASS) {
      const long long idx = base + pos;
      long long r, c;
      if (s.kind == 1) {  // idx = r (r - 1) / 2 + c, r > c
        r = (long long)floor((1.0 + sqrt(1.0 + 8.0 * (double)idx)) * 0.5);
        if (r * (r - 1) / 2 > idx) --r;
        if ((r + 1) * r / 2 <= idx) ++r;
        c = idx - r * (r - 1) / 2;
      } else if (s.kind == 2) {  // idx = r (r + 1) / 2 + c, r >= c
        r = (long long)floor((sqrt(1.0 + 8.0 * (double)idx) - 1.0) * 0.5);
        if (r * (r + 1) / 2 > idx) --r;
        if ((r + 1) * (r + 2) / 2 <= idx) ++r;
        c = idx - r * (r + 1) / 2;
      } else if (s.kind == 3) {  // row r holds its n_a - 1 off-diagonal columns
        r = idx / (s.n_a - 1);
        c = idx - r * (s.n_a - 1);
        if (c >= r) ++c;
      } else {
        r = idx / s.n_b;
        c = idx - r * s.n_b;
      }
      const int vr = order[s.lo_a + (int)r], vc = order[s.lo_b + (int)c];
      er[o + n] = vr;
      ec[o + n] = vc;
      atomicAdd(&deg[vr], 1);
      if (!directed && vr != vc) atomicAdd(&deg[vc], 1);  // (an undirected self-loop is one stored entry)
    }
    ++n;
    ++pos;
    if (pos >= len) break;
  }
  if (!PASS) cnt[ch] = n;
}
__global__ void k_sbm_fill(const int* __restrict__ er, const int* __restrict__ ec, long long m,
                           const int* __restrict__ rowptr, int* __restrict__ cursor, int* __restrict__ col,
                           double* __restrict__ val, int directed) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  const int r = er[e], c = ec[e];
  const int a = rowptr[r] + atomicAdd(&cursor[r], 1);
  col[a] = c;
  val[a] = 1.0;
  if (directed || r == c) return;  // W[r, c] alone / the one entry of a self-loop
  const int b = rowptr[c] + atomicAdd(&cursor[c], 1);
  col[b] = r;
  val[b] = 1.0;
}

}  // namespace gspx

extern "C" int gspx_sbm_build_ex(gspx_ctx* ctx, int64_t N, int k, const int32_t* order, const int64_t* bounds,
                                 const double* M, uint64_t seed, int flags, gspx_knn** out);
extern "C" int gspx_sbm_build(gspx_ctx* ctx, int64_t N, int k, const int32_t* order, const int64_t* bounds,
                              const double* M, uint64_t seed, gspx_knn** out) {
  return gspx_sbm_build_ex(ctx, N, k, order, bounds, M, seed, 0, out);
}

extern "C" int gspx_sbm_build_ex(gspx_ctx* ctx, int64_t N, int k, const int32_t* order, const int64_t* bounds,
                                 const double* M, uint64_t seed, int flags, gspx_knn** out) {
  if (!ctx || !out) return set_err(GSPX_ERR_INVALID, "null ctx or output");
  *out = nullptr;
  if (N < 1 || N >= ((int64_t)1 << 31) - 1 || k < 1 || k > 4096 || !order || !bounds || !M ||
      (flags & ~(GSPX_SBM_DIRECTED | GSPX_SBM_SELF_LOOPS)))
    return set_err(GSPX_ERR_INVALID, "gspx_sbm_build: bad argument");
  const bool directed = (flags & GSPX_SBM_DIRECTED) != 0, loops = (flags & GSPX_SBM_SELF_LOOPS) != 0;
  if (bounds[0] != 0 || bounds[k] != N) return set_err(GSPX_ERR_INVALID, "gspx_sbm_build: bounds must span [0, N]");
  for (int a = 0; a < k; ++a) {
    if (bounds[a + 1] < bounds[a]) return set_err(GSPX_ERR_INVALID, "gspx_sbm_build: bounds must not decrease");
    for (int b = 0; b < k; ++b) {
      const double p = M[(size_t)a * k + b];
      if (!(p >= 0.0 && p <= 1.0)) return set_err(GSPX_ERR_INVALID, "Probabilities should be in [0, 1].");
      if (!directed && p != M[(size_t)b * k + a])
        return set_err(GSPX_ERR_INVALID, "gspx_sbm_build: M must be symmetric (undirected graphs)");
    }
  }
  std::vector<SbmSeg> segs;
  long long nchunks = 0;
  // undirected: the block pairs (a, b <= a), the reference's r >= c half (stochasticblockmodel.py:129, mirrored
  // by utils.symmetrize 'tril'); directed: all k^2 ordered block pairs, entry W[r, c] alone
  for (int a = 0; a < k; ++a)
    for (int b = 0; b <= (directed ? k - 1 : a); ++b) {
      const double p = M[(size_t)a * k + b];
      const long long na = bounds[a + 1] - bounds[a], nbk = bounds[b + 1] - bounds[b];
      int kind = 0;
      long long total = na * nbk;
      if (a == b && !directed) {
        kind = loops ? 2 : 1;
        total = loops ? na * (na + 1) / 2 : na * (na - 1) / 2;
      } else if (a == b && !loops) {
        kind = 3;
        total = na * (na - 1);
      }
      if (p <= 0.0 || total <= 0) continue;
      SbmSeg s{};
      s.total = total;
      s.kind = kind;
      long long c = 64;  // about 16 kept pairs per chunk
      while (c < ((long long)1 << 22) && (double)c * p < 16.0) c <<= 1;
      s.chunk = c;
      s.first = nchunks;
      s.log1mp = std::log1p(-p);
      s.lo_a = (int)bounds[a];
      s.n_a = (int)na;
      s.lo_b = (int)bounds[b];
      s.n_b = (int)nbk;
      nchunks += (total + c - 1) / c;
      segs.push_back(s);
    }
  if (nchunks >= ((long long)1 << 31)) return set_err(GSPX_ERR_INVALID, "gspx_sbm_build: too many candidate chunks");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto t0 = std::chrono::steady_clock::now();
  gspx_knn* h = new gspx_knn();
  h->ctx = ctx;
  h->N = N;
  auto fail = [&](int rc) {
    delete h;
    return rc;
  };
#define KCHK(x)                      \
  do {                               \
    int rc__ = (x);                  \
    if (rc__ != GSPX_OK) return fail(rc__); \
  } while (0)
#define KHIP(x)                                                                            \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) return fail(set_err(GSPX_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__))); \
  } while (0)
  const int n = (int)N;
  DevMem dseg, dorder, cnt, off, er, ec, deg, cursor;
  KCHK(h->rowptr.alloc(((size_t)N + 1) * sizeof(int)));
  KCHK(deg.alloc(((size_t)N + 1) * sizeof(int)));
  KHIP(hipMemsetAsync(deg.p, 0, ((size_t)N + 1) * sizeof(int), st));
  long long m = 0;
  if (nchunks > 0) {
    KCHK(dseg.alloc(segs.size() * sizeof(SbmSeg)));
    KCHK(dorder.alloc((size_t)N * sizeof(int)));
    KCHK(cnt.alloc(((size_t)nchunks + 1) * sizeof(int)));
    KCHK(off.alloc(((size_t)nchunks + 1) * sizeof(int)));
    KHIP(hipMemcpyAsync(dseg.p, segs.data(), segs.size() * sizeof(SbmSeg), hipMemcpyHostToDevice, st));
    KHIP(hipMemcpyAsync(dorder.p, order, (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
    KHIP(hipMemsetAsync(cnt.p, 0, ((size_t)nchunks + 1) * sizeof(int), st));
    const unsigned nbc = (unsigned)((nchunks + 255) / 256);
    hipLaunchKernelGGL((k_sbm_chunks<0>), dim3(nbc), dim3(256), 0, st, dseg.as<SbmSeg>(), (int)segs.size(), nchunks,
                       (unsigned long long)seed, dorder.as<int>(), cnt.as<int>(), (const int*)nullptr,
                       (int*)nullptr, (int*)nullptr, (int*)nullptr, directed ? 1 : 0);
    KCHK(scan_exclusive(ctx, cnt.as<int>(), off.as<int>(), (int)nchunks + 1));
    int mi = 0;
    KHIP(hipMemcpyAsync(&mi, off.as<int>() + nchunks, sizeof(int), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    if (mi < 0 || (long long)mi * 2 >= ((long long)1 << 31))
      return fail(set_err(GSPX_ERR_INVALID, "gspx_sbm_build: more than 2^30 edges"));
    m = mi;
    KCHK(er.alloc((size_t)std::max<long long>(m, 1) * sizeof(int)));
    KCHK(ec.alloc((size_t)std::max<long long>(m, 1) * sizeof(int)));
    hipLaunchKernelGGL((k_sbm_chunks<1>), dim3(nbc), dim3(256), 0, st, dseg.as<SbmSeg>(), (int)segs.size(), nchunks,
                       (unsigned long long)seed, dorder.as<int>(), (int*)nullptr, off.as<int>(), er.as<int>(),
                       ec.as<int>(), deg.as<int>(), directed ? 1 : 0);
  }
  KCHK(scan_exclusive(ctx, deg.as<int>(), h->rowptr.as<int>(), n + 1));
  int stored = 0;  // 2 m for an undirected graph without self-loops; fewer with them, m when directed
  KHIP(hipMemcpyAsync(&stored, h->rowptr.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
  KHIP(hipStreamSynchronize(st));
  h->nnz = stored;
  KCHK(h->col.alloc((size_t)std::max<long long>(stored, 1) * sizeof(int)));
  KCHK(h->val.alloc((size_t)std::max<long long>(stored, 1) * sizeof(double)));
  if (m > 0) {
    KCHK(cursor.alloc(((size_t)N + 1) * sizeof(int)));
    KHIP(hipMemsetAsync(cursor.p, 0, ((size_t)N + 1) * sizeof(int), st));
    hipLaunchKernelGGL(k_sbm_fill, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, er.as<int>(), ec.as<int>(),
                       m, h->rowptr.as<int>(), cursor.as<int>(), h->col.as<int>(), h->val.as<double>(), directed ? 1 : 0);
    hipLaunchKernelGGL(k_knn_row_sort, dim3((n + 255) / 256), dim3(256), 0, st, h->rowptr.as<int>(), n,
                       h->col.as<int>(), h->val.as<double>());
  }
  KHIP(hipGetLastError());
  KHIP(hipStreamSynchronize(st));
#undef KCHK
#undef KHIP
  h->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  return GSPX_OK;
}

// ------------------------------------------------------------------------------------------------
// Radius graphs (NNtype='radius' of NNGraph, nngraph.py:228-287): neighbours = the points within
// epsilon (squared distance <= epsilon^2, the KD-tree's ball query), weights exp(-d^2 / sigma) with
// sigma = mean neighbour distance unless given.  The relation is symmetric and so are the weights, so
// (W + W^T) / 2 = W.  Same grid as the k-NN search with cells of at least epsilon: 3^d cells per query;
// pass 1 counts, pass 2 fills, rows are then sorted by column.
// ------------------------------------------------------------------------------------------------
namespace gspx {

template <int PASS>
__global__ __launch_bounds__(128) void k_radius_query(const double* __restrict__ sorted,
                                                      const int* __restrict__ order,
                                                      const int* __restrict__ start, int N, KnnGrid g,
                                                      double eps2, int* __restrict__ cnt,
                                                      const int* __restrict__ rowptr, int* __restrict__ col,
                                                      double* __restrict__ dist) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int self = order[s];
  double q[3] = {0, 0, 0};
  for (int j = 0; j < g.d; ++j) q[j] = sorted[(size_t)s * g.d + j];
  int c[3];
  knn_cell_of(g, q, c);
  const int r2 = g.d >= 2 ? 1 : 0, r3 = g.d >= 3 ? 1 : 0;
  int n = 0;
  const int o = PASS ? rowptr[self] : 0;
  for (int dz = -r3; dz <= r3; ++dz) {
    const int cz = c[2] + dz;
    if (cz < 0 || cz >= g.n[2]) continue;
    for (int dy = -r2; dy <= r2; ++dy) {
      const int cy = c[1] + dy;
      if (cy < 0 || cy >= g.n[1]) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int cx = c[0] + dx;
        if (cx < 0 || cx >= g.n[0]) continue;
        const int cid = (cz * g.n[1] + cy) * g.n[0] + cx;
        for (int a = start[cid]; a < start[cid + 1]; ++a) {
          const int idx = order[a];
          if (idx == self) continue;
          const double d2 = knn_key(q, sorted + (size_t)a * g.d, g.d, g.metric);
          if (d2 <= eps2) {
            if (PASS) {
              col[o + n] = idx;
              dist[o + n] = g.metric == 0 ? knn_sqrt(d2) : d2;
            }
            ++n;
          }
        }
      }
    }
  }
  if (!PASS) cnt[self] = n;
}
__global__ void k_radius_weights(const double* __restrict__ dist, size_t nnz, double sigma,
                                 double* __restrict__ val) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) {
    const double d = dist[e];
    val[e] = exp(-(d * d) / sigma);
  }
}

}  // namespace gspx

extern "C" int gspx_radius_build(gspx_ctx* ctx, int64_t N, int d, const double* coords, double epsilon,
                                 double sigma, int metric, gspx_knn** out) {
  if (!ctx || !out) return set_err(GSPX_ERR_INVALID, "null ctx or output");
  *out = nullptr;
  if (metric < 0 || metric > 2) return set_err(GSPX_ERR_INVALID, "metric: 0 euclidean, 1 manhattan, 2 max_dist");
  if (N < 1 || N >= ((int64_t)1 << 31) / 64) return set_err(GSPX_ERR_INVALID, "gspx_radius_build: bad N");
  if (d < 1 || d > 64)
    return set_err(GSPX_ERR_INVALID, "gspx_radius_build: the device search covers 1 to 64 dimensions (got %d)", d);
  if (!coords) return set_err(GSPX_ERR_INVALID, "null coordinates");
  if (!(epsilon > 0) || !std::isfinite(epsilon)) return set_err(GSPX_ERR_INVALID, "epsilon must be positive");
  if (!(sigma >= 0) || !std::isfinite(sigma)) return set_err(GSPX_ERR_INVALID, "sigma must be >= 0 (0: mean distance)");
  KnnGrid g{};
  const bool grid_search = d <= 3;  // beyond three dimensions: MFMA candidate sweep + exact test (gspx_knn_bf.hip.h)
  const int dg = grid_search ? d : 0;
  g.d = dg;
  g.metric = metric;
  double hi[3] = {0, 0, 0};
  for (int j = 0; j < 3; ++j) {
    g.lo[j] = 0;
    g.inv_h[j] = 1;
    g.n[j] = 1;
  }
  for (int j = 0; j < dg; ++j) g.lo[j] = hi[j] = coords[j];
  for (int64_t i = 0; i < N; ++i)
    for (int j = 0; j < d; ++j) {
      const double v = coords[i * d + j];
      if (!std::isfinite(v)) return set_err(GSPX_ERR_INVALID, "non-finite coordinate");
      if (j < dg) {
        g.lo[j] = std::min(g.lo[j], v);
        hi[j] = std::max(hi[j], v);
      }
    }
  // cells of at least epsilon (a little more, against rounding at the walls), at most `cap` per axis
  const int cap = d == 1 ? (1 << 21) : (d == 2 ? 2048 : 128);
  int64_t ncells = 1;
  g.h_min = 1e300;
  for (int j = 0; j < dg; ++j) {
    const double ext = hi[j] - g.lo[j];
    int n = ext > 0 ? (int)std::min<double>((double)cap, std::floor(ext / (epsilon * 1.0000001))) : 1;
    n = std::max(n, 1);
    const double h = ext > 0 ? ext / n : 1.0;
    g.n[j] = n;
    g.inv_h[j] = 1.0 / h;
    g.h_min = std::min(g.h_min, h);
    ncells *= n;
  }
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  auto t0 = std::chrono::steady_clock::now();
  gspx_knn* h = new gspx_knn();
  h->ctx = ctx;
  h->N = N;
  h->d = d;
  auto fail = [&](int rc) {
    delete h;
    return rc;
  };
#define KCHK(x)                      \
  do {                               \
    int rc__ = (x);                  \
    if (rc__ != GSPX_OK) return fail(rc__); \
  } while (0)
#define KHIP(x)                                                                            \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) return fail(set_err(GSPX_ERR_HIP, "%s: %s", #x, hipGetErrorString(e__))); \
  } while (0)
  const int n = (int)N;
  DevMem x, sorted, cell, count, start, cursor, order, cnt, partial;
  KCHK(x.alloc((size_t)N * d * sizeof(double)));
  KCHK(sorted.alloc((size_t)N * d * sizeof(double)));
  KCHK(cell.alloc((size_t)N * sizeof(int)));
  KCHK(count.alloc(((size_t)ncells + 1) * sizeof(int)));
  KCHK(start.alloc(((size_t)ncells + 1) * sizeof(int)));
  KCHK(cursor.alloc(((size_t)ncells + 1) * sizeof(int)));
  KCHK(order.alloc((size_t)N * sizeof(int)));
  KCHK(cnt.alloc(((size_t)N + 1) * sizeof(int)));
  KCHK(h->rowptr.alloc(((size_t)N + 1) * sizeof(int)));
  KHIP(hipMemcpyAsync(x.p, coords, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, st));
  KHIP(hipMemsetAsync(count.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
  KHIP(hipMemsetAsync(cursor.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
  KHIP(hipMemsetAsync(cnt.p, 0, ((size_t)N + 1) * sizeof(int), st));
  const int nbN = (n + 255) / 256;
  const double eps2 = metric == 0 ? epsilon * epsilon : epsilon;  // threshold on the comparison key
  DevMem c_off, c_buf;  // candidate lists of the brute-force search
  const bool swept = !grid_search && metric == 0 && N > 256;
  if (!grid_search) {
    KHIP(hipStreamSynchronize(st));
    if (swept) KCHK(radius_candidates(ctx, x.as<double>(), n, d, eps2, c_off, c_buf));
    launch_bf_radius<0>(ctx, x.as<double>(), n, d, metric, eps2, swept ? c_off.as<int>() : nullptr,
                        swept ? c_buf.as<int>() : nullptr, cnt.as<int>(), nullptr, nullptr, nullptr);
  } else {
  hipLaunchKernelGGL(k_knn_cell_count, dim3(nbN), dim3(256), 0, st, x.as<double>(), n, g, cell.as<int>(),
                     count.as<int>());
  KCHK(scan_exclusive(ctx, count.as<int>(), start.as<int>(), (int)ncells + 1));
  hipLaunchKernelGGL(k_knn_scatter, dim3(nbN), dim3(256), 0, st, cell.as<int>(), n, start.as<int>(),
                     cursor.as<int>(), order.as<int>());
  hipLaunchKernelGGL(k_knn_cell_sort, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st,
                     start.as<int>(), (int)ncells, order.as<int>(), x.as<double>(), d, sorted.as<double>());
  hipLaunchKernelGGL((k_radius_query<0>), dim3((n + 127) / 128), dim3(128), 0, st, sorted.as<double>(),
                     order.as<int>(), start.as<int>(), n, g, eps2, cnt.as<int>(), (const int*)nullptr,
                     (int*)nullptr, (double*)nullptr);
  }
  KCHK(scan_exclusive(ctx, cnt.as<int>(), h->rowptr.as<int>(), n + 1));
  int nnz = 0;
  KHIP(hipMemcpyAsync(&nnz, h->rowptr.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
  KHIP(hipStreamSynchronize(st));
  if (nnz < 0) return fail(set_err(GSPX_ERR_INVALID, "gspx_radius_build: more than 2^31 neighbour pairs"));
  h->nnz = nnz;
  KCHK(h->col.alloc((size_t)std::max(nnz, 1) * sizeof(int)));
  KCHK(h->val.alloc((size_t)std::max(nnz, 1) * sizeof(double)));
  KCHK(h->dist.alloc((size_t)std::max(nnz, 1) * sizeof(double)));
  if (nnz > 0) {
    if (!grid_search)
      launch_bf_radius<1>(ctx, x.as<double>(), n, d, metric, eps2, swept ? c_off.as<int>() : nullptr,
                          swept ? c_buf.as<int>() : nullptr, nullptr, h->rowptr.as<int>(), h->col.as<int>(),
                          h->dist.as<double>());
    else
    hipLaunchKernelGGL((k_radius_query<1>), dim3((n + 127) / 128), dim3(128), 0, st, sorted.as<double>(),
                       order.as<int>(), start.as<int>(), n, g, eps2, (int*)nullptr, h->rowptr.as<int>(),
                       h->col.as<int>(), h->dist.as<double>());
    hipLaunchKernelGGL(k_knn_row_sort, dim3(nbN), dim3(256), 0, st, h->rowptr.as<int>(), n, h->col.as<int>(),
                       h->dist.as<double>());
    if (sigma == 0.0) {  // mean neighbour distance (nngraph.py:248-262)
      const int nb = 1024;
      KCHK(partial.alloc((size_t)nb * sizeof(double)));
      hipLaunchKernelGGL(k_knn_sum_partial, dim3(nb), dim3(256), 0, st, h->dist.as<double>(), (size_t)nnz,
                         partial.as<double>());
      std::vector<double> hp(nb);
      KHIP(hipMemcpyAsync(hp.data(), partial.p, nb * sizeof(double), hipMemcpyDeviceToHost, st));
      KHIP(hipStreamSynchronize(st));
      double s = 0;
      for (double v : hp) s += v;
      sigma = s / (double)nnz;
      if (!(sigma > 0)) return fail(set_err(GSPX_ERR_INVALID, "gspx_radius_build: all neighbour distances are zero"));
    }
    hipLaunchKernelGGL(k_radius_weights, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st,
                       h->dist.as<double>(), (size_t)nnz, sigma, h->val.as<double>());
  } else if (sigma == 0.0) {
    return fail(set_err(GSPX_ERR_INVALID, "No neighbors found"));  // nngraph.py:263-264
  }
  h->sigma = sigma;
  KHIP(hipGetLastError());
  KHIP(hipStreamSynchronize(st));
#undef KCHK
#undef KHIP
  h->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  return GSPX_OK;
}
