// gspx_ops_kernels.hip.h - device kernels of the operators that reuse the engine's CSR next to the
// Chebyshev path: L x, Dirichlet energy, the conjugate-gradient
// loop of Tikhonov regression, gradient / divergence.  gfx950 only.
//
// Reference call sites:
//   Graph.dirichlet_energy          pygsp/graphs/graph.py:642-702      x.T.dot(L.dot(x))
//   compute_differential_operator   pygsp/graphs/difference.py:26-166  D (incidence matrix)
//   grad / div                      pygsp/graphs/difference.py:168-331 D.T.dot(x), D.dot(y)
//   regression_tikhonov (tau > 0)   pygsp/learning.py:324-337          scipy.sparse.linalg.cg
#pragma once

#include "gspx_kernels.hip.h"

namespace gspx {

// A = s * L + diag(m) on the internal layout (same pattern as L: every row owns a diagonal slot).
// m is indexed by internal row; null = no diagonal term.
template <typename T>
__global__ void k_affine_values(const int* __restrict__ rptr, const int* __restrict__ rcol,
                                const T* __restrict__ rval, int N, T s, const T* __restrict__ m,
                                T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const T mi = m ? m[i] : T(0);
  for (int j = rptr[i] & ~3; j < (rptr[i + 1] & ~3); ++j) {
    const int c = rcol[j];
    T v = s * rval[j];
    if (c == i) v += mi;
    out[j] = (c == N) ? T(0) : v;
  }
}

// ---- column-wise reductions over N x ld row-major panels ------------------------------------------
// partial[b][c] = sum over the block's rows of A[i][c] * B[i][c]  (double accumulation, fixed order).
// ldp = power of two >= ld (<= 256): thread t works on column t % ldp, rows t / ldp + k * (256 / ldp).
template <typename T>
__global__ __launch_bounds__(256) void k_coldot_partial(const T* __restrict__ A, const T* __restrict__ B,
                                                        int N, int ld, int ldp,
                                                        double* __restrict__ partial) {
  __shared__ double ws[256];
  const int c = threadIdx.x & (ldp - 1);
  const int r0 = threadIdx.x / ldp;
  const int rstep = 256 / ldp;
  double acc = 0;
  if (c < ld) {
    const size_t stride = (size_t)gridDim.x * rstep;
    size_t i = (size_t)blockIdx.x * rstep + r0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;  // four rows in flight per thread
    for (; i + 3 * stride < (size_t)N; i += 4 * stride) {
      a0 += (double)A[i * ld + c] * (double)B[i * ld + c];
      a1 += (double)A[(i + stride) * ld + c] * (double)B[(i + stride) * ld + c];
      a2 += (double)A[(i + 2 * stride) * ld + c] * (double)B[(i + 2 * stride) * ld + c];
      a3 += (double)A[(i + 3 * stride) * ld + c] * (double)B[(i + 3 * stride) * ld + c];
    }
    for (; i < (size_t)N; i += stride) a0 += (double)A[i * ld + c] * (double)B[i * ld + c];
    acc = (a0 + a1) + (a2 + a3);
  }
  ws[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < ldp) {
    double s = 0;
    for (int k = 0; k < rstep; ++k) s += ws[k * ldp + threadIdx.x];
    if ((int)threadIdx.x < ld) partial[(size_t)blockIdx.x * ld + threadIdx.x] = s;
  }
}
// out[c] = sum_b partial[b][c]: one 64-lane wave per column, fixed summation tree (deterministic)
__global__ __launch_bounds__(64) void k_colsum(const double* __restrict__ partial, int nb, int ld,
                                               double* __restrict__ out) {
  const int c = blockIdx.x;
  if (c >= ld) return;
  double s = 0;
  for (int b = threadIdx.x; b < nb; b += 64) s += partial[(size_t)b * ld + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (threadIdx.x == 0) out[c] = s;
}

// Gram block on the matrix cores: G[a0 + a][c0 + c] += sum_i X[i][a0 + a] * Y[i][c0 + c] for a 64 x 64 block
// of the Gram matrix (na, nc <= 64 valid columns).  This IS a dense panel contraction (8 flop/byte at 64
// signals), so it runs on MFMA: v_mfma_f64_16x16x4f64 takes A = X^T (16 signals x 4 rows) and B = Y
// (4 rows x 16 signals) straight from coalesced row loads - lane l holds row l / 16, column l % 16 of
// both - and accumulates a 16 x 16 tile; a wave keeps the 4 x 4 tiles of its block in registers and
// walks the rows four at a time.  fp32 panels are converted on load: the sums are double either way.
// partial[w][a][c] (ld x ld per wave, only this block's entries written).
template <typename T>
__global__ __launch_bounds__(256) void k_gram_mfma(const T* __restrict__ X, const T* __restrict__ Y, int N,
                                                   int ld, int a0, int na, int c0, int nc,
                                                   double* __restrict__ partial) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int kq = lane >> 4, cq = lane & 15;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (long r4 = gw; r4 * 4 < (long)N; r4 += nw) {
    const long row = r4 * 4 + kq;
    const bool rok = row < (long)N;
    double xa[4], yb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ca = t * 16 + cq;
      xa[t] = (rok && ca < na) ? (double)X[(size_t)row * ld + a0 + ca] : 0.0;
      yb[t] = (rok && ca < nc) ? (double)Y[(size_t)row * ld + c0 + ca] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i], yb[j], acc[i][j], 0, 0, 0);
  }
  // D layout of the f64 16x16x4 instruction: lane l holds rows (l / 16) + 4 e (e = 0..3), column l % 16
  double* out = partial + (size_t)gw * ld * ld;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int aa = i * 16 + kq + 4 * e, cc = j * 16 + cq;
        if (aa < na && cc < nc) out[(size_t)(a0 + aa) * ld + (c0 + cc)] = acc[i][j][e];
      }
}

// ---- conjugate gradient, one independent system per column (scipy.sparse.linalg.cg's recurrence) ---
struct CgScalars {  // per column, device resident
  double* rho_prev;
  double* rho_cur;
  double* pq;
  double* atol;     // max(atol, rtol * ||b||)
  double* alpha;
  double* beta;
  int* active;      // 1 while the column iterates
  int* iters;
  int* any_active;  // [1]
};

// start of an iteration (scipy _isolve.cg: "if bnrm2 ... if norm(r) < atol: return"): freeze the
// columns that have converged, beta = rho_cur / rho_prev for the others (0 on the first iteration)
__global__ void k_cg_pre(CgScalars s, const double* __restrict__ rr, int ld, int first) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  int act = s.active[c];
  if (act && sqrt(rr[c]) < s.atol[c]) act = 0;
  s.active[c] = act;
  if (act) {
    s.rho_cur[c] = rr[c];
    s.beta[c] = first ? 0.0 : rr[c] / s.rho_prev[c];
    s.iters[c] += 1;
    atomicOr(s.any_active, 1);
  }
}
// p = r + beta p on the active columns
// (element indices fit 32 bits: a panel is below 2 GiB; the column comes from a 32-bit modulo)
template <typename T>
__global__ void k_cg_p(const T* __restrict__ r, T* __restrict__ p, size_t total, int ld, CgScalars s) {
  const unsigned n = (unsigned)total, step = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int c = (int)(i % (unsigned)ld);
    if (s.active[c]) p[i] = r[i] + (T)s.beta[c] * p[i];
  }
}
__global__ void k_cg_post(CgScalars s, const double* __restrict__ pq, int ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  if (s.active[c]) {
    s.alpha[c] = s.rho_cur[c] / pq[c];
    s.rho_prev[c] = s.rho_cur[c];
  }
}
// x += alpha p; r -= alpha q on the active columns
template <typename T>
__global__ void k_cg_xr(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                        const T* __restrict__ q, size_t total, int ld, CgScalars s) {
  const unsigned n = (unsigned)total, step = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int c = (int)(i % (unsigned)ld);
    if (s.active[c]) {
      const T al = (T)s.alpha[c];
      x[i] += al * p[i];
      r[i] -= al * q[i];
    }
  }
}
// The same update with ||r_new||^2 per column accumulated on the way (the next iteration's convergence test and
// rho): thread layout, row order and summation order are k_coldot_partial's, so the sums equal a separate
// coldot(r, r) bit for bit - and the panel r is not read a second time (14U -> 13U per iteration).
template <typename T>
__global__ __launch_bounds__(256) void k_cg_xr_dot(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                                                   const T* __restrict__ q, int N, int ld, int ldp, CgScalars s,
                                                   double* __restrict__ partial) {
  __shared__ double ws[256];
  const int c = threadIdx.x & (ldp - 1);
  const int r0 = threadIdx.x / ldp;
  const int rstep = 256 / ldp;
  double acc = 0;
  if (c < ld) {
    const bool on = s.active[c] != 0;
    const T al = on ? (T)s.alpha[c] : T(0);
    auto upd = [&](size_t i) {
      const size_t e = i * ld + c;
      T rv = r[e];
      if (on) {
        x[e] += al * p[e];
        rv -= al * q[e];
        r[e] = rv;
      }
      return (double)rv * (double)rv;
    };
    const size_t stride = (size_t)gridDim.x * rstep;
    size_t i = (size_t)blockIdx.x * rstep + r0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (; i + 3 * stride < (size_t)N; i += 4 * stride) {
      a0 += upd(i);
      a1 += upd(i + stride);
      a2 += upd(i + 2 * stride);
      a3 += upd(i + 3 * stride);
    }
    for (; i < (size_t)N; i += stride) a0 += upd(i);
    acc = (a0 + a1) + (a2 + a3);
  }
  ws[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < ldp) {
    double t = 0;
    for (int k = 0; k < rstep; ++k) t += ws[k * ldp + threadIdx.x];
    if ((int)threadIdx.x < ld) partial[(size_t)blockIdx.x * ld + threadIdx.x] = t;
  }
}
__global__ void k_cg_init(CgScalars s, const double* __restrict__ bb, int ld, double rtol, double atol) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  const double t = rtol * sqrt(bb[c]);
  s.atol[c] = t > atol ? t : atol;
  s.active[c] = bb[c] > 0.0 ? 1 : 0;  // scipy returns x = 0 at once for a zero right-hand side
  s.iters[c] = 0;
  s.rho_prev[c] = 1.0;
}
// out[i][c] = m[i] * y[i][c]  (panels in the same order as m)
template <typename T>
__global__ void k_rowscale(const T* __restrict__ m, const T* __restrict__ y, T* __restrict__ out,
                           size_t total, int ld) {
  const unsigned n = (unsigned)total, step = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    out[i] = m[i / (unsigned)ld] * y[i];
}

// ---- differential operator of an undirected graph (difference.py:140-166) --------------------------
// Edges = stored entries (i, j) of the canonical Laplacian with j > i, in row-major order - the order
// of sparse.triu(W, format='coo') that Graph.get_edge_list returns (graph.py:1019-1029).
__global__ void k_edge_count(const int* __restrict__ lptr, const int* __restrict__ lcol, int N,
                             int* __restrict__ up, int* __restrict__ low) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int u = 0, l = 0;
  for (int j = lptr[i]; j < lptr[i + 1]; ++j) {
    u += lcol[j] > i;
    l += lcol[j] < i;
  }
  up[i] = u;
  low[i] = l;
}
// src, dst, D values at the source (negative) and the target (positive) of every edge, its weight,
// and for every vertex the list of edges that end in it (tedge, via the lower-triangular entries)
template <typename T>
__global__ void k_edge_fill(const int* __restrict__ lptr, const int* __restrict__ lcol,
                            const T* __restrict__ lval, const T* __restrict__ dw, int N, int lap_type,
                            const int* __restrict__ eoff, const int* __restrict__ toff,
                            int* __restrict__ esrc, int* __restrict__ edst, T* __restrict__ cs,
                            T* __restrict__ ct, T* __restrict__ ew, int* __restrict__ tedge) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int e = eoff[i], t = toff[i];
  for (int j = lptr[i]; j < lptr[i + 1]; ++j) {
    const int c = lcol[j];
    if (c > i) {
      esrc[e] = i;
      edst[e] = c;
      if (lap_type == 0) {  // combinatorial: W_ij = -L_ij
        const T w = -lval[j];
        ew[e] = w;
        cs[e] = -sqrt(w);
        ct[e] = sqrt(w);
      } else {              // normalized: L_ij = -W_ij / (sqrt(d_i) sqrt(d_j))
        const T w = -lval[j] * sqrt(dw[i]) * sqrt(dw[c]);
        ew[e] = w;
        cs[e] = -sqrt(w / dw[i]);
        ct[e] = sqrt(w / dw[c]);
      }
      ++e;
    } else if (c < i) {
      // edge (c, i): its id = eoff[c] + rank of i among the columns > c of row c (sorted)
      int lo = lptr[c], hi = lptr[c + 1];
      int first_up = lo;
      {  // first entry of row c with column > c
        int a = lo, b = hi;
        while (a < b) {
          const int mid = (a + b) >> 1;
          if (lcol[mid] > c) b = mid; else a = mid + 1;
        }
        first_up = a;
      }
      int a = first_up, b = hi;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (lcol[mid] >= i) b = mid; else a = mid + 1;
      }
      tedge[t++] = eoff[c] + (a - first_up);
    }
  }
}
// D values of a caller's edge list (difference.py:151-161): -sqrt(w) / +sqrt(w), or -sqrt(w / d_source) /
// +sqrt(w / d_target) for the normalized Laplacian, both divided by sqrt(2) on a directed graph
template <typename T>
__global__ void k_edge_values(const int* __restrict__ esrc, const int* __restrict__ edst, const T* __restrict__ ew,
                              const T* __restrict__ dw, int E, int lap_type, int directed, T* __restrict__ cs,
                              T* __restrict__ ct) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  const T w = ew[k];
  T a, b;
  if (lap_type == 0) {
    a = -sqrt(w);
    b = -a;
  } else {
    a = -sqrt(w / dw[esrc[k]]);
    b = sqrt(w / dw[edst[k]]);
  }
  if (directed) {
    const T root2 = sqrt((T)2);
    a /= root2;
    b /= root2;
  }
  cs[k] = a;
  ct[k] = b;
}
// grad: y[k][:] = cs[k] * x[src[k]][:] + ct[k] * x[dst[k]][:]      (D.T.dot(x), difference.py:244)
// 2-D thread blocks (cw columns x 256/cw edges): consecutive lanes walk the signals of one edge, no
// index division anywhere
template <typename T>
__global__ __launch_bounds__(256) void k_grad(const int* __restrict__ esrc, const int* __restrict__ edst,
                                              const T* __restrict__ cs, const T* __restrict__ ct,
                                              const T* __restrict__ x, T* __restrict__ y, size_t E, int ld,
                                              int cw) {
  const int cx = threadIdx.x % cw, ry = threadIdx.x / cw, rpb = 256 / cw;
  for (size_t k = (size_t)blockIdx.x * rpb + ry; k < E; k += (size_t)gridDim.x * rpb) {
    const size_t s = (size_t)esrc[k] * ld, d = (size_t)edst[k] * ld;
    const T a = cs[k], b = ct[k];
    for (int c = cx; c < ld; c += cw) y[k * ld + c] = a * x[s + c] + b * x[d + c];
  }
}
// div: z[i][:] = sum_{k: src = i} cs[k] y[k][:] + sum_{k: dst = i} ct[k] y[k][:]   (D.dot(y), :331)
template <typename T>
__global__ __launch_bounds__(256) void k_div(const int* __restrict__ eoff, const int* __restrict__ toff,
                                             const int* __restrict__ tedge, const T* __restrict__ cs,
                                             const T* __restrict__ ct, const T* __restrict__ y,
                                             T* __restrict__ z, int N, int ld, int cw) {
  const int cx = threadIdx.x % cw, ry = threadIdx.x / cw, rpb = 256 / cw;
  for (size_t i = (size_t)blockIdx.x * rpb + ry; i < (size_t)N; i += (size_t)gridDim.x * rpb) {
    const int e0 = eoff[i], e1 = eoff[i + 1], t0 = toff[i], t1 = toff[i + 1];
    for (int c = cx; c < ld; c += cw) {
      T acc = 0;
      // the order of scipy's csc_matvec over D's rows does not matter for the value beyond rounding
      for (int k = e0; k < e1; ++k) acc += cs[k] * y[(size_t)k * ld + c];
      for (int m = t0; m < t1; ++m) {
        const int k = tedge[m];
        acc += ct[k] * y[(size_t)k * ld + c];
      }
      z[i * ld + c] = acc;
    }
  }
}

// The same two operators walking the VERTICES in the engine's internal (space-filling-curve) order, one
// lane group per vertex, 16-byte lanes, XCD-contiguous vertex ranges: the rows of the other endpoints
// (grad) and the edge rows shared by two vertices (div) were touched a moment ago by a neighbouring
// vertex on the same XCD and come from its L2, where the edge-order kernels above fetch them from HBM
// again.  Edges of a vertex are consecutive (the list is sorted by source), so grad writes whole runs.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_grad_v(const int* __restrict__ perm, const int* __restrict__ eoff,
                                                const int* __restrict__ edst, const T* __restrict__ cs,
                                                const T* __restrict__ ct, const T* __restrict__ x,
                                                T* __restrict__ y, int N, int ld, int gs, int per_xcd) {
  typedef typename VT<T, VEC>::t V;
  const int lane = threadIdx.x % gs, grp = threadIdx.x / gs, gpb = 256 / gs, cpr = ld / VEC;
  const int lo = (int)(blockIdx.x & 7) * per_xcd, hi = min(N, lo + per_xcd);
  const int stride = (int)(gridDim.x >> 3) * gpb;
  for (int i = lo + (int)(blockIdx.x >> 3) * gpb + grp; i < hi; i += stride) {
    const int v = perm ? perm[i] : i;
    const int e0 = eoff[v], e1 = eoff[v + 1];
    for (int c = lane; c < cpr; c += gs) {
      const V xv = *(const V*)(x + (size_t)v * ld + (size_t)c * VEC);
      for (int k = e0; k < e1; k += 4) {  // four neighbour rows in flight per lane
        int kk[4];
        V xd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kk[j] = k + j < e1 ? k + j : e1 - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) xd[j] = *(const V*)(x + (size_t)edst[kk[j]] * ld + (size_t)c * VEC);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < e1) *(V*)(y + (size_t)kk[j] * ld + (size_t)c * VEC) = cs[kk[j]] * xv + ct[kk[j]] * xd[j];
      }
    }
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_div_v(const int* __restrict__ perm, const int* __restrict__ eoff,
                                               const int* __restrict__ toff, const int* __restrict__ tedge,
                                               const T* __restrict__ cs, const T* __restrict__ ct,
                                               const T* __restrict__ y, T* __restrict__ z, int N, int ld,
                                               int gs, int per_xcd) {
  typedef typename VT<T, VEC>::t V;
  const int lane = threadIdx.x % gs, grp = threadIdx.x / gs, gpb = 256 / gs, cpr = ld / VEC;
  const int lo = (int)(blockIdx.x & 7) * per_xcd, hi = min(N, lo + per_xcd);
  const int stride = (int)(gridDim.x >> 3) * gpb;
  for (int i = lo + (int)(blockIdx.x >> 3) * gpb + grp; i < hi; i += stride) {
    const int v = perm ? perm[i] : i;
    const int e0 = eoff[v], e1 = eoff[v + 1], t0 = toff[v], t1 = toff[v + 1];
    for (int c = lane; c < cpr; c += gs) {
      V acc = 0;  // same order as k_div: the vertex's own edges, then the edges pointing at it
      for (int k = e0; k < e1; k += 4) {  // four edge rows in flight per lane
        V yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = *(const V*)(y + (size_t)(k + j < e1 ? k + j : e1 - 1) * ld + (size_t)c * VEC);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < e1) acc += cs[k + j] * yv[j];
      }
      for (int m = t0; m < t1; m += 4) {
        int kk[4];
        V yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kk[j] = tedge[m + j < t1 ? m + j : t1 - 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = *(const V*)(y + (size_t)kk[j] * ld + (size_t)c * VEC);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (m + j < t1) acc += ct[kk[j]] * yv[j];
      }
      *(V*)(z + (size_t)v * ld + (size_t)c * VEC) = acc;
    }
  }
}

}  // namespace gspx
