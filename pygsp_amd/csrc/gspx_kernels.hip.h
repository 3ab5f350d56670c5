// gspx_kernels.hip.h — gfx950 (MI355X / CDNA4) device code of libgspx.
//
// Written for wave64 CDNA4 only: no CUDA shims, no multi-backend paths.
//
// What is computed (reference: pygsp/filters/approximations.py:93-112):
//     a1 = a2 = lmax/2
//     F      = (2/a1) * (L - a2*I)                     ("factor", approximations.py:105)
//     T_0    = x ;  T_1 = 0.5 * F T_0                  (== (L x - a2 x)/a1, approximations.py:99)
//     T_k    = F T_{k-1} - T_{k-2}                     (approximations.py:107)
//     r_f    = 0.5 c_f0 T_0 + sum_{k>=1} c_fk T_k      (approximations.py:103,109)
//
// Data layout in HBM:
//   * signals: row-major [N][ld] panels (ld = number of signals in the batch), so one vertex's
//     signals are contiguous: a neighbour gather is one coalesced ld*elt-byte read.
//   * matrix: "padded CSR" in the engine's internal vertex order: every row owns a diagonal
//     slot, row length is a multiple of 4, pad entries are (col = N, val = 0).  col = N makes
//     the gather address land exactly at num_records of the buffer descriptor, so the hardware
//     bounds check returns 0 without touching memory: no predicates, no branches in the gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gspx {

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#ifndef GSPX_STREAM_AUX
#define GSPX_STREAM_AUX 0  // cache-policy bits of the streaming T_{k-2} loads / T_k stores (2 = nt)
#endif
#define GSPX_POISON 0x80000000u  // voffset >= num_records for every panel we accept (< 2 GiB)

// ---------------------------------------------------------------------------------------------
// vector-of-T helpers: VEC elements per lane, at most 16 bytes
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC> struct VT;

template <> struct VT<float, 1> {
  typedef float t;
  static __device__ __forceinline__ t bload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
  }
  static __device__ __forceinline__ t sload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, GSPX_STREAM_AUX));
  }
  static __device__ __forceinline__ void sstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, v), r, o, 0, GSPX_STREAM_AUX);
  }
  static __device__ __forceinline__ void bstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, v), r, o, 0, 0);
  }
};
template <> struct VT<float, 2> {
  typedef float t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ t bload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0));
  }
  static __device__ __forceinline__ void bstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, o, 0, 0);
  }
  static __device__ __forceinline__ t sload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, GSPX_STREAM_AUX));
  }
  static __device__ __forceinline__ void sstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, o, 0, GSPX_STREAM_AUX);
  }
};
template <> struct VT<float, 4> {
  typedef float t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ t bload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
  }
  static __device__ __forceinline__ void bstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, 0);
  }
  static __device__ __forceinline__ t sload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, GSPX_STREAM_AUX));
  }
  static __device__ __forceinline__ void sstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, GSPX_STREAM_AUX);
  }
};
template <> struct VT<double, 1> {
  typedef double t;
  static __device__ __forceinline__ t bload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0));
  }
  static __device__ __forceinline__ void bstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, o, 0, 0);
  }
  static __device__ __forceinline__ t sload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, GSPX_STREAM_AUX));
  }
  static __device__ __forceinline__ void sstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, o, 0, GSPX_STREAM_AUX);
  }
};
template <> struct VT<double, 2> {
  typedef double t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ t bload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
  }
  static __device__ __forceinline__ void bstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, 0);
  }
  static __device__ __forceinline__ t sload(rsrc_t r, u32 o) {
    return __builtin_bit_cast(t, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, GSPX_STREAM_AUX));
  }
  static __device__ __forceinline__ void sstore(rsrc_t r, u32 o, t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, GSPX_STREAM_AUX);
  }
};

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int off) { return __shfl_xor(v, off); }

template <typename T, int VEC>
__device__ __forceinline__ typename VT<T, VEC>::t vec_shfl_xor(typename VT<T, VEC>::t v, int off) {
  if constexpr (VEC == 1) {
    return __shfl_xor(v, off);
  } else {
    typename VT<T, VEC>::t o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = __shfl_xor(v[k], off);
    return o;
  }
}

// ---------------------------------------------------------------------------------------------
// arguments of one recurrence step:  out = scale * (F cur) + gamma * old   (+ optional flush)
// ---------------------------------------------------------------------------------------------
template <typename T> struct StepArgs {
  const int* rowptr;  // [N+1] padded-CSR row starts (internal vertex order)
  const int* col;     // [nnz_int + 64]
  const T* val;       // [nnz_int + 64] factor values F
  const T* cur;       // T_{k-1}   [N][ld]
  const T* old;       // T_{k-2}   [N][ld]   (may alias out; unused when gamma == 0)
  T* out;             // T_k       [N][ld]
  int N;
  u32 ld;        // signals in this batch (elements per row)
  u32 curbytes;  // N*ld*sizeof(T)  (< 2 GiB)
  T scale, gamma;
  T beta;        // coefficient of T_{k-1}'s own row (Newton-form steps; 0 for the recurrence)
  // flush: r_f (=|+=) w_new*T_k + w_cur*T_{k-1} + w_old*T_{k-2}
  int flush;     // 0 none, 1 write, 2 accumulate
  int final;     // 1: write to y (caller's vertex order, row stride ldy) instead of racc
  int nf;
  int nin;       // MODE 2 only: number of extra input panels summed into the row (synthesis
                 // by Clenshaw: out += sum_f wts[f] * racc[f][row]); 0 otherwise
  const T* wts;  // device, [nf][3] (flush) or [nin] (extra inputs)
  T* racc;       // [nf][N][ld] internal accumulators
  T* y;          // [nf][N][ldy] (already offset to the batch's first column)
  u32 ldy;
  const int* perm;  // internal row -> caller's row (nullable = identity)
  // row -> workgroup mapping
  int rows_per_wave;
  int wpb;          // waves per workgroup (panel kernel: 4, 8 or 16)
  int lds_pad;      // host only: extra dynamic LDS bytes of the launch (occupancy experiments)
  int nchunks;      // number of (wpb*rows_per_wave)-row chunks
  int cpx;          // chunks per XCD (xcd_remap) or 0 for plain order
  int reverse;      // 1: sweep the rows from the end (alternate steps: the tail of the previous
                    // step's output is still in the Infinity Cache)
};

// ---------------------------------------------------------------------------------------------
// PANEL kernel.  W = 2^WLOG2 lanes span the signals of one vertex (VEC each); a wave works on
// R = 64/W consecutive rows at a time ("row set"), lane group r owning row r of the set.
//
//  * Row metadata (rowptr / col / val) is wave-uniform per row, so it travels through the
//    scalar cache (s_load_dwordx4/x8) and never touches the vector memory pipeline; lane groups
//    pick their row's entry with v_cndmask.  The vector pipeline carries only the neighbour
//    gathers (one coalesced buffer_load per stored entry per row), T_{k-2} and T_k.
//  * Gathers are issued 8 deep per lane (two 4-entry chunks) before the first FMA.  Lanes whose
//    row has run out of chunks get an out-of-range offset: the descriptor bounds check returns 0
//    and generates no memory traffic, so there is no divergent control flow in the loop.
//  * vmcnt retires in order, so the streaming T_{k-2} read of the NEXT row set is issued right
//    after the current set's first gathers: it is younger than them (does not delay their
//    wait) and has a whole row-set's latency to arrive.
// ---------------------------------------------------------------------------------------------
// Lane group r of a wave picks its row's value out of R wave-uniform (SGPR) values.  The pick
// is a bitwise blend  a ^ ((a ^ b) & m)  with per-lane masks m in {0, ~0} that are made opaque
// to the optimiser once per kernel (LaneSel): spelled as selects, LLVM rewrites "select among
// the R members" into a dynamically indexed private array and the AMDGPU backend parks that in
// LDS/scratch.  (a ^ b) of two SGPRs is scalar work; the blend costs 2 VALU ops per dword.
struct LaneSel {
  u32 m0, m1;  // ~0 where bit 0 / bit 1 of the lane's row index is set
};
__device__ __forceinline__ LaneSel make_lane_sel(int r) {
  LaneSel s;
  s.m0 = (u32) - (r & 1);
  s.m1 = (u32) - ((r >> 1) & 1);
  asm volatile("" : "+v"(s.m0), "+v"(s.m1));
  return s;
}
__device__ __forceinline__ u32 blend(u32 a, u32 b, u32 m) { return a ^ ((a ^ b) & m); }
__device__ __forceinline__ int blend(int a, int b, u32 m) { return (int)blend((u32)a, (u32)b, m); }
__device__ __forceinline__ float blend(float a, float b, u32 m) {
  return __builtin_bit_cast(float, blend(__builtin_bit_cast(u32, a), __builtin_bit_cast(u32, b), m));
}
__device__ __forceinline__ double blend(double a, double b, u32 m) {
  const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
  u32x2 o;
  o.x = blend(ua.x, ub.x, m);
  o.y = blend(ua.y, ub.y, m);
  return __builtin_bit_cast(double, o);
}

template <typename T, int R> struct RowVals;
template <typename T> struct RowVals<T, 1> {
  T a;
  template <class F> static __device__ __forceinline__ RowVals make(F f) { return RowVals{f(0)}; }
  __device__ __forceinline__ T at(int) const { return a; }
  __device__ __forceinline__ T pick(const LaneSel&) const { return a; }
};
template <typename T> struct RowVals<T, 2> {
  T a, b;
  template <class F> static __device__ __forceinline__ RowVals make(F f) {
    return RowVals{f(0), f(1)};
  }
  __device__ __forceinline__ T at(int q) const { return q == 0 ? a : b; }
  __device__ __forceinline__ T pick(const LaneSel& s) const { return blend(a, b, s.m0); }
};
template <typename T> struct RowVals<T, 4> {
  T a, b, c, d;
  template <class F> static __device__ __forceinline__ RowVals make(F f) {
    return RowVals{f(0), f(1), f(2), f(3)};
  }
  __device__ __forceinline__ T at(int q) const { return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d)); }
  __device__ __forceinline__ T pick(const LaneSel& s) const {
    return blend(blend(a, b, s.m0), blend(c, d, s.m0), s.m1);
  }
};

// per-wave constants of the panel kernel
template <typename T, int VEC> struct PanelCtx {
  const int* __restrict__ rowptr;
  const int* __restrict__ col;
  const T* __restrict__ val;
  const T* __restrict__ wts;
  const int* __restrict__ perm;
  rsrc_t rc;    // T_{k-1} panel (gathers)
  rsrc_t rold;  // T_{k-2} panel
  rsrc_t rout;  // T_k panel (may be the same memory as rold)
  rsrc_t rra;   // accumulator plane 0 (flush kernels)
  u32 ldb, lane_off, colel;
  int r;          // this lane's row within the set
  LaneSel sel;
  int row0;       // first row of this wave
  int set_stride; // rows between consecutive sets of this wave
  int nsets;
  bool lane_on;
};

// One row set.  `ov_use`/`ra_use` were prefetched by the previous set; `ov_pf`/`ra_pf` receive the
// prefetch for the next one (distinct registers: the caller alternates two pairs, so no copy -
// and therefore no wait on the in-flight prefetch - is needed at the top of the loop).
// MODE: 0 plain step, 1 step + flush, 2 plain step with the beta * T_{k-1}[row] term
template <typename T, int VEC, int R, int MODE>
__device__ __forceinline__ bool panel_row_set(const PanelCtx<T, VEC>& c, const StepArgs<T>& a,
                                              const int it,
                                              const typename VT<T, VEC>::t& ov_use,
                                              const typename VT<T, VEC>::t& ra_use,
                                              typename VT<T, VEC>::t& ov_pf,
                                              typename VT<T, VEC>::t& ra_pf) {
  typedef VT<T, VEC> X;
  typedef typename X::t V;
  constexpr bool FLUSH = MODE == 1;
  constexpr bool SELF = MODE != 0;
  const int srow = __builtin_amdgcn_readfirstlane(c.row0 + it * c.set_stride);
  if (srow >= a.N) return false;
  // rowptr is padded past N with the total entry count, so rows >= N read as empty
  // (low 2 bits of a rowptr entry carry that row's pad count; starts are multiples of 4)
  const RowVals<int, R> rs = RowVals<int, R>::make([&](int q) { return c.rowptr[srow + q] & ~3; });
  const RowVals<int, R> nch = RowVals<int, R>::make(
      [&](int q) { return (c.rowptr[srow + q + 1] >> 2) - (c.rowptr[srow + q] >> 2); });
  const RowVals<int, R> last =
      RowVals<int, R>::make([&](int q) { return nch.at(q) > 0 ? nch.at(q) - 1 : 0; });
  int nmax = nch.at(0);
  if constexpr (R >= 2) nmax = nch.at(1) > nmax ? nch.at(1) : nmax;
  if constexpr (R >= 4) {
    nmax = nch.at(2) > nmax ? nch.at(2) : nmax;
    nmax = nch.at(3) > nmax ? nch.at(3) : nmax;
  }
  const int nch_l = nch.pick(c.sel);
  const int myrow = srow + c.r;
  const bool row_on = myrow < a.N && c.lane_on;

  V x0, x1, x2, x3, x4, x5, x6, x7;
  T v0, v1, v2, v3, v4, v5, v6, v7;
  V acc = 0;
  V curv = 0;

  // one gather: entry u of chunk kk of every row of the set
  // `pm` is 0 for lanes whose row still has chunk kk and GSPX_POISON otherwise; it is opaque to the
  // optimiser so that the (scalar) col/val loads stay unconditional and batched.
  auto gather = [&](const RowVals<int, R>& base, const u32 pm, const int u, V& xo, T& vo) {
    const RowVals<int, R> cs = RowVals<int, R>::make([&](int q) { return c.col[base.at(q) + u]; });
    const RowVals<T, R> vs = RowVals<T, R>::make([&](int q) { return c.val[base.at(q) + u]; });
    const int cc = cs.pick(c.sel);
    vo = vs.pick(c.sel);
    xo = X::bload(c.rc, ((u32)cc * c.ldb + c.lane_off) | pm);
  };
  // issue the 8 gathers of chunks k, k+1 (no FMA yet)
  auto issue_pair = [&](const int k) {
    {
      const RowVals<int, R> base = RowVals<int, R>::make(
          [&](int q) { return rs.at(q) + 4 * (k < last.at(q) ? k : last.at(q)); });
      u32 pm = k < nch_l ? 0u : GSPX_POISON;
      asm volatile("" : "+v"(pm));
      gather(base, pm, 0, x0, v0);
      gather(base, pm, 1, x1, v1);
      gather(base, pm, 2, x2, v2);
      gather(base, pm, 3, x3, v3);
    }
    {
      const int k1 = k + 1;
      const RowVals<int, R> base = RowVals<int, R>::make(
          [&](int q) { return rs.at(q) + 4 * (k1 < last.at(q) ? k1 : last.at(q)); });
      u32 pm = k1 < nch_l ? 0u : GSPX_POISON;
      asm volatile("" : "+v"(pm));
      gather(base, pm, 0, x4, v4);
      gather(base, pm, 1, x5, v5);
      gather(base, pm, 2, x6, v6);
      gather(base, pm, 3, x7, v7);
    }
  };
  auto fma_pair = [&]() {
    acc += v0 * x0;
    acc += v1 * x1;
    acc += v2 * x2;
    acc += v3 * x3;
    acc += v4 * x4;
    acc += v5 * x5;
    acc += v6 * x6;
    acc += v7 * x7;
  };

  // all chunk pairs but the last.  Entry 0 of every row is its diagonal slot, so the first
  // gather of the first pair IS T_{k-1}[row].
  int k = 0;
  for (; k + 2 < nmax; k += 2) {
    issue_pair(k);
    fma_pair();
    if constexpr (SELF) {
      if (k == 0) curv = x0;
    }
  }
  // last pair (every row owns a diagonal slot, so nmax >= 1), then the streaming prefetch for the
  // next row set.  The prefetch is issued AFTER this set's last gathers: vmcnt retires in order,
  // so being younger it never delays their wait, and it is unconditional (clamped address) so
  // that the compiler's vmcnt bookkeeping is exact and the final FMA does not wait for it.
  issue_pair(k);
  {
    const int nrow = myrow + c.set_stride;
    const bool pf_on = (it + 1 < c.nsets) && nrow < a.N;
    const u32 po = pf_on ? (u32)nrow * c.ldb + c.lane_off : GSPX_POISON;
    ov_pf = X::sload(c.rold, po);
    if constexpr (FLUSH) ra_pf = X::sload(c.rra, po);
  }
  fma_pair();
  if constexpr (SELF) {
    if (k == 0) curv = x0;
  }

  // T_k = scale * (F T_{k-1}) + gamma * T_{k-2}; lanes without a row store out of range (dropped)
  V nv = a.scale * acc;
  nv += a.gamma * ov_use;  // gamma == 0: the host points `old` at `cur`, the product vanishes
  if constexpr (SELF) nv += a.beta * curv;
  if constexpr (MODE == 2) {
    if (a.nin > 0 || a.final) {
      // extra input panels (vector-coefficient Clenshaw) and/or a final store in caller order
      if (row_on) {
        const size_t o = (size_t)myrow * a.ld + c.colel;
        const size_t plane_r = (size_t)a.N * a.ld;
        for (int f = 0; f < a.nin; ++f) nv += c.wts[f] * *(const V*)(a.racc + f * plane_r + o);
        if (a.final) {
          const size_t orow = c.perm ? (size_t)c.perm[myrow] : (size_t)myrow;
          *(V*)(a.y + orow * a.ldy + c.colel) = nv;
        }
      }
      if (a.final) return true;
    }
  }
  X::sstore(c.rout, row_on ? (u32)myrow * c.ldb + c.lane_off : GSPX_POISON, nv);

  if constexpr (FLUSH) {
    if (row_on) {
      const size_t o = (size_t)myrow * a.ld + c.colel;
      size_t orow = (size_t)myrow;
      if (a.final && c.perm) orow = (size_t)c.perm[myrow];
      const size_t plane_r = (size_t)a.N * a.ld;
      const size_t plane_y = (size_t)a.N * a.ldy;
      for (int f = 0; f < a.nf; ++f) {
        const T wn = c.wts[3 * f + 0], wc = c.wts[3 * f + 1], wo = c.wts[3 * f + 2];
        V res = wn * nv + wc * curv + wo * ov_use;
        if (a.flush == 2) res += (f == 0) ? ra_use : *(const V*)(a.racc + f * plane_r + o);
        if (a.final)
          *(V*)(a.y + f * plane_y + orow * a.ldy + c.colel) = res;
        else
          *(V*)(a.racc + f * plane_r + o) = res;
      }
    }
  }
  return true;
}

template <typename T, int VEC, int WLOG2, int MODE>
__global__ __launch_bounds__(1024) void k_step_panel(const int* __restrict__ rowptr,
                                                    const int* __restrict__ col,
                                                    const T* __restrict__ val,
                                                    const T* __restrict__ cur,
                                                    const T* __restrict__ wts,
                                                    const int* __restrict__ perm,
                                                    const StepArgs<T> a) {
  typedef typename VT<T, VEC>::t V;
  constexpr int W = 1 << WLOG2;
  constexpr int R = 64 / W;
  constexpr bool FLUSH = MODE == 1;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = lane & (W - 1);

  int chunk = blockIdx.x;
  if (a.cpx > 0) chunk = (chunk & 7) * a.cpx + (chunk >> 3);  // contiguous row range per XCD
  if (chunk >= a.nchunks) return;
  if (a.reverse) chunk = a.nchunks - 1 - chunk;

  PanelCtx<T, VEC> c;
  c.rowptr = rowptr;
  c.col = col;
  c.val = val;
  c.wts = wts;
  c.perm = perm;
  c.r = lane >> WLOG2;
  c.sel = make_lane_sel(c.r);
  c.colel = (blockIdx.y * W + w) * VEC;  // first signal handled by this lane
  c.lane_on = c.colel < a.ld;
  c.ldb = a.ld * (u32)sizeof(T);
  c.lane_off = c.lane_on ? c.colel * (u32)sizeof(T) : GSPX_POISON;
  c.rc = __builtin_amdgcn_make_buffer_rsrc((void*)cur, 0, a.curbytes, 0x00020000);
  c.rold = __builtin_amdgcn_make_buffer_rsrc((void*)a.old, 0, a.curbytes, 0x00020000);
  c.rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.curbytes, 0x00020000);
  c.rra = __builtin_amdgcn_make_buffer_rsrc((void*)a.racc, 0, a.curbytes, 0x00020000);
  c.row0 = (chunk * a.wpb + wave) * a.rows_per_wave;  // each wave owns a contiguous run of rows
  c.set_stride = R;
  c.nsets = a.rows_per_wave / R;

  if (c.row0 >= a.N) return;
  V ovA = 0, raA = 0, ovB = 0, raB = 0;
  {
    typedef VT<T, VEC> X;
    const int myrow = c.row0 + c.r;
    const u32 po = myrow < a.N ? (u32)myrow * c.ldb + c.lane_off : GSPX_POISON;
    ovA = X::sload(c.rold, po);
    if constexpr (FLUSH) raA = X::sload(c.rra, po);
  }
#pragma unroll 1
  for (int it = 0; it < c.nsets; it += 2) {
    if (!panel_row_set<T, VEC, R, MODE>(c, a, it, ovA, raA, ovB, raB)) break;
    if (it + 1 >= c.nsets) break;
    if (!panel_row_set<T, VEC, R, MODE>(c, a, it + 1, ovB, raB, ovA, raA)) break;
  }
}

// ---------------------------------------------------------------------------------------------
// byte offsets col*ldb of the stored entries for one panel width (cached on the graph; read by the
// LDS-staged kernel below).  Pads get 0x80000000: the gather fails the descriptor's bounds check once the
// lane offset is added - no traffic.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_coff(const int* __restrict__ rptr, const int* __restrict__ rcol, int N, u32 ldb,
                       u32* __restrict__ coff) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int j = rptr[i] & ~3; j < (rptr[i + 1] & ~3); ++j) {
    const int c = rcol[j];
    coff[j] = (c == N) ? GSPX_POISON : (u32)c * ldb;
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-STAGED panel kernel.  Measured on MI355X (profiles/): the gather pipeline moves ~80 B/clk/CU
// with 16-byte-per-lane loads but only ~40 with 4-8-byte ones, and a wave that fetches its row
// metadata through dependent global round trips spends ~6 us per row.  So:
//   * lanes: W = 2^WLOG2 lanes x 16 bytes span one row, R = 64/W rows of a "row set" share every
//     gather instruction (1 KiB per instruction);
//   * the CSR slice of ALL rows of a wave (byte offsets + factor values, contiguous because the
//     rows are consecutive) is copied once, coalesced, into a wave-private LDS region; lane group r
//     then reads its row's 4-entry chunk with ds_read_b128 (broadcast inside the group) - no
//     scalar loads, no cross-lane blends, no per-row rowptr fetch (the wave's rowptr block sits in
//     one VGPR and is read with ds_bpermute / v_readlane);
//   * per stored entry the wave issues 1/R gather + (1 + VEC)/R VALU instructions;
//   * pads carry offset 0x80000000: with the lane offset added they fail the descriptor's bounds
//     check and cost no memory traffic.
// ---------------------------------------------------------------------------------------------
#define GSPX_LDS_CAP 512  // CSR entries staged per wave

template <typename T, int VEC> struct LdsCtx {
  typedef T T4 __attribute__((ext_vector_type(4)));
  const u32x4* s_coff;  // this wave's LDS slice (offsets)
  const T4* s_val;      // this wave's LDS slice (values)
  const u32* __restrict__ coff;
  const T* __restrict__ val;
  const T* __restrict__ wts;
  const int* __restrict__ perm;
  rsrc_t rc, rold, rout, rra;
  u32 ldb, lane_off, colel;
  int r, row0, nsets, v_rp, seg_s;
  bool lane_on;
};

// One row set.  ov_use/ra_use were requested by the previous set, ov_pf/ra_pf receive the request
// for the next one (the caller alternates two register pairs: no copy, hence no wait, at the top).
template <typename T, int VEC, int R, int MODE, bool STAGED>
__device__ __forceinline__ void lds_row_set(const LdsCtx<T, VEC>& c, const StepArgs<T>& a, const int j,
                                            const typename VT<T, VEC>::t& ov_use,
                                            const typename VT<T, VEC>::t& ra_use,
                                            typename VT<T, VEC>::t& ov_pf,
                                            typename VT<T, VEC>::t& ra_pf) {
  typedef VT<T, VEC> X;
  typedef typename X::t V;
  typedef T T4 __attribute__((ext_vector_type(4)));
  constexpr bool FLUSH = MODE == 1;
  constexpr bool SELF = MODE != 0;
  const int srow = c.row0 + j * R;
  const int myrow = srow + c.r;
  const bool row_on = myrow < a.N && c.lane_on;
  // this lane group's row bounds, and the set's longest row (wave-uniform loop bound)
  const int my0 = __shfl(c.v_rp, j * R + c.r) & ~3;
  const int my1 = __shfl(c.v_rp, j * R + c.r + 1) & ~3;
  const int my_nch = (my1 - my0) >> 2;
  int nmax = 0;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int n = ((__builtin_amdgcn_readlane(c.v_rp, j * R + q + 1) & ~3) -
                   (__builtin_amdgcn_readlane(c.v_rp, j * R + q) & ~3)) >> 2;
    nmax = n > nmax ? n : nmax;
  }
  const int my_q = (my0 - c.seg_s) >> 2;  // first chunk of my row inside the staged slice

  // Consume the previous set's streaming request HERE (it is a whole set old): the wait sits at
  // the top of the set, and nothing below depends on a T_{k-2} load any more - otherwise the
  // compiler guards the epilogue with vmcnt(0), which would also wait for the request issued
  // for the NEXT set.
  V acc = a.gamma * ov_use;  // gamma == 0: the host points `old` at `cur`, the product vanishes
  V rbase = 0;
  if constexpr (FLUSH) {
    rbase = c.wts[2] * ov_use;  // w_old of filter 0 (the common single-filter case)
    if (a.flush == 2) rbase += ra_use;
  }
  asm volatile("" : "+v"(acc), "+v"(rbase));
  V sum = 0;
  V curv = 0;

  // the streaming request for the next set, valid only on the last pass (else out of range)
  const int nrow = myrow + R;
  const bool pf_on = (j + 1 < c.nsets) && nrow < a.N;
  const u32 pf_off = pf_on ? (u32)nrow * c.ldb + c.lane_off : GSPX_POISON;

  for (int k = 0; k < nmax; k += 2) {
    // two chunks (8 entries) per pass: offsets from LDS, 8 gathers in flight, then the values
    const bool on0 = k < my_nch, on1 = k + 1 < my_nch;
    u32x4 c0, c1;
    if constexpr (STAGED) {
      c0 = c.s_coff[on0 ? my_q + k : 0];
      c1 = c.s_coff[on1 ? my_q + k + 1 : 0];
    } else {
      c0 = *(const u32x4*)(c.coff + my0 + 4 * (on0 ? k : 0));
      c1 = *(const u32x4*)(c.coff + my0 + 4 * (on1 ? k + 1 : 0));
    }
    const u32 p0 = on0 ? c.lane_off : GSPX_POISON, p1 = on1 ? c.lane_off : GSPX_POISON;
    const V x0 = X::bload(c.rc, c0.x + p0), x1 = X::bload(c.rc, c0.y + p0);
    const V x2 = X::bload(c.rc, c0.z + p0), x3 = X::bload(c.rc, c0.w + p0);
    const V x4 = X::bload(c.rc, c1.x + p1), x5 = X::bload(c.rc, c1.y + p1);
    const V x6 = X::bload(c.rc, c1.z + p1), x7 = X::bload(c.rc, c1.w + p1);
    // (unconditional: a pass that is not the last one asks for an out-of-range address, which
    // costs an issue slot but no memory traffic and keeps the vmcnt bookkeeping exact)
    const u32 po = (k + 2 >= nmax) ? pf_off : GSPX_POISON;
    __builtin_amdgcn_sched_barrier(0);  // gathers first: the streaming request must be younger
    ov_pf = X::sload(c.rold, po);
    if constexpr (FLUSH) ra_pf = X::sload(c.rra, po);
    __builtin_amdgcn_sched_barrier(0);  // the requests above stay above the waits below
    T4 v0, v1;
    if constexpr (STAGED) {
      v0 = c.s_val[on0 ? my_q + k : 0];
      v1 = c.s_val[on1 ? my_q + k + 1 : 0];
    } else {
      v0 = *(const T4*)(c.val + my0 + 4 * (on0 ? k : 0));
      v1 = *(const T4*)(c.val + my0 + 4 * (on1 ? k + 1 : 0));
    }
    sum += v0.x * x0; sum += v0.y * x1; sum += v0.z * x2; sum += v0.w * x3;
    sum += v1.x * x4; sum += v1.y * x5; sum += v1.z * x6; sum += v1.w * x7;
    if constexpr (SELF) {
      if (k == 0) curv = x0;  // entry 0 of every row is its diagonal slot: x0 = T_{k-1}[row]
    }
  }

  V nv = a.scale * sum + acc;
  if constexpr (SELF) nv += a.beta * curv;
  if constexpr (MODE == 2) {
    if (a.nin > 0 || a.final) {
      if (row_on) {
        const size_t o = (size_t)myrow * a.ld + c.colel;
        const size_t plane_r = (size_t)a.N * a.ld;
        for (int f = 0; f < a.nin; ++f) nv += c.wts[f] * *(const V*)(a.racc + f * plane_r + o);
        if (a.final) {
          const size_t orow = c.perm ? (size_t)c.perm[myrow] : (size_t)myrow;
          *(V*)(a.y + orow * a.ldy + c.colel) = nv;
        }
      }
      if (a.final) return;
    }
  }
  X::sstore(c.rout, row_on ? (u32)myrow * c.ldb + c.lane_off : GSPX_POISON, nv);
  if constexpr (FLUSH) {
    if (row_on) {
      const size_t o = (size_t)myrow * a.ld + c.colel;
      size_t orow = (size_t)myrow;
      if (a.final && c.perm) orow = (size_t)c.perm[myrow];
      const size_t plane_r = (size_t)a.N * a.ld;
      const size_t plane_y = (size_t)a.N * a.ldy;
      for (int f = 0; f < a.nf; ++f) {
        const T wn = c.wts[3 * f + 0], wc = c.wts[3 * f + 1], wo = c.wts[3 * f + 2];
        V res = wn * nv + wc * curv;
        if (f == 0) {
          res += rbase;
        } else {
          res += wo * ov_use;
          if (a.flush == 2) res += *(const V*)(a.racc + f * plane_r + o);
        }
        if (a.final)
          *(V*)(a.y + f * plane_y + orow * a.ldy + c.colel) = res;
        else
          *(V*)(a.racc + f * plane_r + o) = res;
      }
    }
  }
}

template <typename T, int VEC, int R, int MODE, bool STAGED>
__device__ __forceinline__ void lds_run_sets(const LdsCtx<T, VEC>& c, const StepArgs<T>& a) {
  typedef VT<T, VEC> X;
  typedef typename X::t V;
  constexpr bool FLUSH = MODE == 1;
  V ovA, raA = 0, ovB = 0, raB = 0;
  {
    const int myrow = c.row0 + c.r;
    const u32 po = myrow < a.N ? (u32)myrow * c.ldb + c.lane_off : GSPX_POISON;
    ovA = X::sload(c.rold, po);
    if constexpr (FLUSH) raA = X::sload(c.rra, po);
  }
#pragma unroll 1
  for (int j = 0; j < c.nsets; j += 2) {
    if (c.row0 + j * R >= a.N) break;
    lds_row_set<T, VEC, R, MODE, STAGED>(c, a, j, ovA, raA, ovB, raB);
    if (j + 1 >= c.nsets || c.row0 + (j + 1) * R >= a.N) break;
    lds_row_set<T, VEC, R, MODE, STAGED>(c, a, j + 1, ovB, raB, ovA, raA);
  }
}

template <typename T, int VEC, int WLOG2, int MODE>
__global__ __launch_bounds__(256) void k_step_lds(const int* __restrict__ rowptr,
                                                  const u32* __restrict__ coff,
                                                  const T* __restrict__ val,
                                                  const T* __restrict__ cur,
                                                  const T* __restrict__ wts,
                                                  const int* __restrict__ perm,
                                                  const StepArgs<T> a) {
  constexpr int W = 1 << WLOG2;
  constexpr int R = 64 / W;
  typedef T T4 __attribute__((ext_vector_type(4)));
  __shared__ u32x4 s_coff[4][GSPX_LDS_CAP / 4];
  __shared__ T4 s_val[4][GSPX_LDS_CAP / 4];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = lane & (W - 1);

  int chunk = blockIdx.x;
  if (a.cpx > 0) chunk = (chunk & 7) * a.cpx + (chunk >> 3);
  if (chunk >= a.nchunks) return;
  if (a.reverse) chunk = a.nchunks - 1 - chunk;

  LdsCtx<T, VEC> c;
  c.s_coff = s_coff[wave];
  c.s_val = s_val[wave];
  c.coff = coff;
  c.val = val;
  c.wts = wts;
  c.perm = perm;
  c.r = lane >> WLOG2;
  c.colel = (blockIdx.y * W + w) * VEC;
  c.lane_on = c.colel < a.ld;
  c.ldb = a.ld * (u32)sizeof(T);
  c.lane_off = c.lane_on ? c.colel * (u32)sizeof(T) : GSPX_POISON;
  c.rc = __builtin_amdgcn_make_buffer_rsrc((void*)cur, 0, a.curbytes, 0x00020000);
  c.rold = __builtin_amdgcn_make_buffer_rsrc((void*)a.old, 0, a.curbytes, 0x00020000);
  c.rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.curbytes, 0x00020000);
  c.rra = __builtin_amdgcn_make_buffer_rsrc((void*)a.racc, 0, a.curbytes, 0x00020000);
  c.row0 = (chunk * 4 + wave) * a.rows_per_wave;
  if (c.row0 >= a.N) return;
  c.nsets = a.rows_per_wave / R;  // rows_per_wave is a multiple of R, at most 32

  // rowptr block of this wave: lane l holds rowptr[row0 + l] (rowptr is padded past N)
  c.v_rp = rowptr[(c.row0 + lane) < a.N ? (c.row0 + lane) : a.N];
  c.seg_s = __builtin_amdgcn_readfirstlane(c.v_rp) & ~3;
  const int seg_e = __builtin_amdgcn_readlane(c.v_rp, a.rows_per_wave) & ~3;
  const int seg_n = seg_e - c.seg_s;  // stored entries of the wave's rows (multiple of 4)

  if (seg_n <= GSPX_LDS_CAP) {
    // stage offsets + values: lane l copies chunks l, l + 64 (coalesced 16/32-byte loads)
    for (int q = lane; q * 4 < seg_n; q += 64) {
      s_coff[wave][q] = *(const u32x4*)(coff + c.seg_s + 4 * q);
      s_val[wave][q] = *(const T4*)(val + c.seg_s + 4 * q);
    }
    lds_run_sets<T, VEC, R, MODE, true>(c, a);
  } else {
    lds_run_sets<T, VEC, R, MODE, false>(c, a);  // very long rows: metadata straight from L2
  }
}

// ---------------------------------------------------------------------------------------------
// Row products from LDS (k_step_tile; the two-level experimental kernels under experimental/ use it too)
// ---------------------------------------------------------------------------------------------
typedef unsigned short u16;
typedef unsigned char u8;

// Dot product of one padded CSR row with rows of an LDS tile, the entries in LDS too (pads resolved), one 4-entry
// chunk per trip: small register footprint, the staged kernels keep a prefetched block's row lists live across it.
template <typename T, typename V, int LG = 16, typename I = u16>  // LG: lanes (16-byte pieces) per tile row;
__device__ __forceinline__ V lds_row_dot(const T* val, const I* idx, int len, const V* tile, int lane16,
                                         V& self) {                   // I: tile positions, 16 or 8 bits
  typedef T T4 __attribute__((ext_vector_type(4)));
  typedef I I4 __attribute__((ext_vector_type(4)));
  V acc = 0;
  self = 0;
  for (int j = 0; j < len; j += 4) {
    const I4 ia = *(const I4*)(idx + j);
    const T4 va = *(const T4*)(val + j);
    const V t0 = tile[ia.x * LG + lane16];
    const V t1 = tile[ia.y * LG + lane16];
    const V t2 = tile[ia.z * LG + lane16];
    const V t3 = tile[ia.w * LG + lane16];
    if (j == 0) self = t0;
    acc += va.x * t0;
    acc += va.y * t1;
    acc += va.z * t2;
    acc += va.w * t3;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// NARROW kernel (1..4 signals): several rows per wave; W lanes span the signals, G lanes split a
// row's entries (per-lane vector loads of col/val, coalesced because a wave's rows are
// consecutive in the padded CSR), xor-shuffle reduction across the G lanes.
// ---------------------------------------------------------------------------------------------
template <typename T, bool FLUSH>
__global__ __launch_bounds__(256) void k_step_narrow(const StepArgs<T> a, const int wlog2,
                                                     const int glog2) {
  typedef VT<T, 1> X;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int W = 1 << wlog2, G = 1 << glog2;
  const int w = lane & (W - 1);
  const int g = (lane >> wlog2) & (G - 1);
  const int rlog2 = 6 - wlog2 - glog2;  // rows per wave = 2^rlog2
  const int r_in_wave = lane >> (wlog2 + glog2);

  int chunk = blockIdx.x;
  if (a.cpx > 0) chunk = (chunk & 7) * a.cpx + (chunk >> 3);
  if (chunk >= a.nchunks) return;

  const bool lane_on = (u32)w < a.ld;
  const u32 ldb = a.ld * (u32)sizeof(T);
  const u32 lane_off = lane_on ? (u32)w * (u32)sizeof(T) : GSPX_POISON;
  const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)a.cur, 0, a.curbytes, 0x00020000);

  const int rows_per_pass = 4 << rlog2;  // rows a workgroup covers per pass
  const int row_begin = chunk * a.rows_per_wave * rows_per_pass;
#pragma unroll 1
  for (int i = 0; i < a.rows_per_wave; ++i) {
    const int row = row_begin + i * rows_per_pass + (wave << rlog2) + r_in_wave;
    const bool row_on = row < a.N;
    int s = 0, e = 0;
    if (row_on) {
      s = a.rowptr[row] & ~3;  // low 2 bits carry the row's pad count
      e = a.rowptr[row + 1] & ~3;
    }
    T acc = 0;
    for (int j = s + g; j < e; j += 2 * G) {
      const int c0 = a.col[j];
      const T v0 = a.val[j];
      const int j1 = j + G;
      const bool on1 = j1 < e;
      const int c1 = on1 ? a.col[j1] : a.N;
      const T v1 = on1 ? a.val[j1] : T(0);
      const T x0 = X::bload(rc, (u32)c0 * ldb + lane_off);
      const T x1 = X::bload(rc, (u32)c1 * ldb + lane_off);
      acc += v0 * x0;
      acc += v1 * x1;
    }
    for (int off = W; off < (W << glog2); off <<= 1) acc += __shfl_xor(acc, off);

    if (row_on && g == 0 && lane_on) {
      const size_t o = (size_t)row * a.ld + w;
      T nv = a.scale * acc;
      T ov = 0;
      if (a.gamma != T(0)) {
        ov = a.old[o];
        nv += a.gamma * ov;
      }
      if (a.beta != T(0)) nv += a.beta * a.cur[o];
      if (!FLUSH && (a.nin > 0 || a.final)) {
        const size_t plane_in = (size_t)a.N * a.ld;
        for (int f = 0; f < a.nin; ++f) nv += a.wts[f] * a.racc[f * plane_in + o];
        if (a.final) {
          const size_t orow2 = a.perm ? (size_t)a.perm[row] : (size_t)row;
          a.y[orow2 * a.ldy + w] = nv;
          continue;
        }
      }
      a.out[o] = nv;
      if constexpr (FLUSH) {
        const T curv = a.cur[o];
        size_t orow = (size_t)row;
        if (a.final && a.perm) orow = (size_t)a.perm[row];
        const size_t plane_r = (size_t)a.N * a.ld;
        const size_t plane_y = (size_t)a.N * a.ldy;
        for (int f = 0; f < a.nf; ++f) {
          const T wn = a.wts[3 * f + 0], wc = a.wts[3 * f + 1], wo = a.wts[3 * f + 2];
          T r = wn * nv + wc * curv + wo * ov;
          if (a.flush == 2) r += a.racc[f * plane_r + o];
          if (a.final)
            a.y[f * plane_y + orow * a.ldy + w] = r;
          else
            a.racc[f * plane_r + o] = r;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Deferred combine (filterbanks): all T_k of the batch are kept ([K+1][N][ld] slots) and the
// Nf outputs are formed in ONE streaming pass:  y_f = sum_k c'_fk T_k.   This is the
// tall-skinny contraction [N*ld x (K+1)] . [(K+1) x Nf]; at ~3 flop/byte it is HBM-bound, so
// plain VALU FMAs (not MFMA) are the right tool.  NFB filters are accumulated per pass.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int NFB>
__global__ __launch_bounds__(256) void k_combine(const T* __restrict__ slots, int nslots,
                                                 size_t slot_stride, const T* __restrict__ cf,
                                                 int M, int f0, int nf_here, int N, u32 ld, T* y,
                                                 u32 ldy, size_t plane_y, const int* __restrict__ perm,
                                                 int accumulate, u32 pitch) {  // pitch: slot row pitch (>= ld)
  typedef typename VT<T, VEC>::t V;
  const u32 cpr = ld / VEC;  // vector chunks per row
  const size_t total = (size_t)N * cpr;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * 256) {
    const u32 row = (u32)(idx / cpr);
    const u32 ch = (u32)(idx - (size_t)row * cpr);
    const size_t o = (size_t)row * pitch + (size_t)ch * VEC;
    V acc[NFB];
#pragma unroll
    for (int f = 0; f < NFB; ++f) acc[f] = 0;
    for (int k = 0; k < nslots; ++k) {
      const V t = *(const V*)(slots + (size_t)k * slot_stride + o);
#pragma unroll
      for (int f = 0; f < NFB; ++f)
        if (f < nf_here) acc[f] += cf[(size_t)(f0 + f) * M + k] * t;
    }
    const size_t orow = perm ? (size_t)perm[row] : (size_t)row;
#pragma unroll
    for (int f = 0; f < NFB; ++f)
      if (f < nf_here) {
        V* dst = (V*)(y + (size_t)(f0 + f) * plane_y + orow * ldy + (size_t)ch * VEC);
        if (accumulate)
          *dst = *dst + acc[f];
        else
          *dst = acc[f];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// permute-in: T_0[i][c] = x[perm[i]][c0 + c]   (also the plain copy when perm == null)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_permute_in(const T* __restrict__ x, u32 ldx,
                                                    T* __restrict__ out, u32 ld, int N,
                                                    const int* __restrict__ perm) {
  typedef typename VT<T, VEC>::t V;
  const u32 cpr = ld / VEC;
  const size_t total = (size_t)N * cpr;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * 256) {
    const u32 row = (u32)(idx / cpr);
    const u32 ch = (u32)(idx - (size_t)row * cpr);
    const size_t src = perm ? (size_t)perm[row] : (size_t)row;
    *(V*)(out + (size_t)row * ld + (size_t)ch * VEC) =
        *(const V*)(x + src * ldx + (size_t)ch * VEC);
  }
}

// Panels whose rows are not made of 16-byte pieces (an odd number of fp64 signals, fp32 signals not in fours, a
// misaligned y) are worked on with padded rows, so that they take the tile kernels too:
//   in:  T_0[i][c] = c < ld ? x[perm[i]][c] : 0      (row pitch ldw of the work panel)
//   out: y[perm[i]][c] = r[i][c], c < ld
template <typename T>
__global__ __launch_bounds__(256) void k_permute_in_pad(const T* __restrict__ x, u32 ldx, T* __restrict__ out,
                                                        u32 ldw, u32 ld, int N, const int* __restrict__ perm) {
  const size_t total = (size_t)N * ldw;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const u32 row = (u32)(idx / ldw);
    const u32 c = (u32)(idx - (size_t)row * ldw);
    const size_t src = perm ? (size_t)perm[row] : (size_t)row;
    out[idx] = c < ld ? x[src * ldx + c] : T(0);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_permute_out_pad(const T* __restrict__ r, u32 ldw, T* __restrict__ y, u32 ldy,
                                                         u32 ld, int N, const int* __restrict__ perm) {
  const size_t total = (size_t)N * ld;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const u32 row = (u32)(idx / ld);
    const u32 c = (u32)(idx - (size_t)row * ld);
    const size_t dst = perm ? (size_t)perm[row] : (size_t)row;
    y[dst * ldy + c] = r[(size_t)row * ldw + c];
  }
}

// ---------------------------------------------------------------------------------------------
// graph build kernels (replace graph.py:618-628, 830-838)
// ---------------------------------------------------------------------------------------------
// dw[i] = sum_j W_ij, sequential in ascending column order: for an exactly symmetric W this is
// the same addition order as scipy's column sums W.sum(axis=0) (graph.py:833).
template <typename T>
__global__ void k_degree(const int* __restrict__ ptr, const T* __restrict__ val, int N,
                         T* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  T s = 0;
  for (int j = ptr[i]; j < ptr[i + 1]; ++j) s += val[j];
  dw[i] = s;
}

// The ingredients of Graph._get_upper_bound (graph.py:933-960) in one pass over W, per 256-row block:
// part[4 b + 0..3] = max W_ij, max (dw_i + dw_j) over stored entries, max (dw_i + (W dw)_i / dw_i), number of
// zero-degree rows (their 0 / 0 makes numpy's maximum NaN: the host layer then drops that candidate, as
// Python's min() does).  The row sums run in column order without fused multiply-add, like scipy's W.dot(dw).
__global__ __launch_bounds__(256) void k_lmax_bounds(const int* __restrict__ ptr, const int* __restrict__ col,
                                                     const double* __restrict__ val, const double* __restrict__ dw,
                                                     int N, double* __restrict__ part) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  double wmax = -1e300, emax = -1e300, mmax = -1e300, zero = 0;
  if (i < N) {
    const double di = dw[i];
    double s = 0;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
      const double w = val[j], dj = dw[col[j]];
      const double p = w * dj;
      s = s + p;
      wmax = fmax(wmax, w);
      emax = fmax(emax, di + dj);
    }
    if (di == 0.0) zero = 1;
    else mmax = di + s / di;
  }
  __shared__ double sh[4][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    wmax = fmax(wmax, __shfl_down(wmax, off));
    emax = fmax(emax, __shfl_down(emax, off));
    mmax = fmax(mmax, __shfl_down(mmax, off));
    zero += __shfl_down(zero, off);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[wv][0] = wmax;
    sh[wv][1] = emax;
    sh[wv][2] = mmax;
    sh[wv][3] = zero;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = part + (size_t)blockIdx.x * 4;
    o[0] = fmax(fmax(sh[0][0], sh[1][0]), fmax(sh[2][0], sh[3][0]));
    o[1] = fmax(fmax(sh[0][1], sh[1][1]), fmax(sh[2][1], sh[3][1]));
    o[2] = fmax(fmax(sh[0][2], sh[1][2]), fmax(sh[2][2], sh[3][2]));
    o[3] = sh[0][3] + sh[1][3] + sh[2][3] + sh[3][3];
  }
}

// d^{-1/2} with the reference's isolated-vertex rule (graph.py:622-624)
template <typename T> __device__ __forceinline__ T inv_sqrt_deg(T dw) {
  return dw == T(0) ? T(0) : T(1) / sqrt(dw);
}

// value of L_ij for an off-diagonal stored W_ij
template <typename T>
__device__ __forceinline__ T lap_offdiag(int lap_type, T w, T di, T dj) {
  if (lap_type == 0) return -w;
  return -((di * w) * dj);  // (D*W)*D, graph.py:626
}
// value of L_ii given dw_i and the (possibly absent) self-loop weight
template <typename T> __device__ __forceinline__ T lap_diag(int lap_type, T dw, T wii, T di) {
  if (lap_type == 0) return dw - wii;
  if (dw == T(0)) return T(0);   // L[disconnected, disconnected] = 0, graph.py:627
  return T(1) - (di * wii) * di;
}

// pass 1 (count) / pass 2 (fill) of canonical L = D - W  or  I - D^-1/2 W D^-1/2, zeros dropped
template <typename T, bool FILL>
__global__ void k_lap_build(const int* __restrict__ wptr, const int* __restrict__ wcol,
                            const T* __restrict__ wval, const T* __restrict__ dw, int N,
                            int lap_type, int* __restrict__ cnt, const int* __restrict__ lptr,
                            int* __restrict__ lcol, T* __restrict__ lval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const T dwi = dw[i];
  const T di = lap_type == 1 ? inv_sqrt_deg(dwi) : T(0);
  // self loop weight
  T wii = 0;
  for (int j = wptr[i]; j < wptr[i + 1]; ++j)
    if (wcol[j] == i) wii = wval[j];
  const T dval = lap_diag(lap_type, dwi, wii, di);
  int n = 0;
  int o = FILL ? lptr[i] : 0;
  bool diag_done = false;
  for (int j = wptr[i]; j < wptr[i + 1]; ++j) {
    const int c = wcol[j];
    if (c == i) continue;
    if (!diag_done && c > i) {
      diag_done = true;
      if (dval != T(0)) {
        if (FILL) { lcol[o] = i; lval[o] = dval; ++o; }
        ++n;
      }
    }
    const T dj = lap_type == 1 ? inv_sqrt_deg(dw[c]) : T(0);
    const T v = lap_offdiag(lap_type, wval[j], di, dj);
    if (v != T(0)) {
      if (FILL) { lcol[o] = c; lval[o] = v; ++o; }
      ++n;
    }
  }
  if (!diag_done && dval != T(0)) {
    if (FILL) { lcol[o] = i; lval[o] = dval; ++o; }
    ++n;
  }
  if (!FILL) cnt[i] = n;
}

__global__ void k_inverse_perm(const int* __restrict__ perm, int N, int* __restrict__ iperm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) iperm[perm[i]] = i;
}

// canonical L -> internal padded CSR.  pass 1: padded row lengths; pass 2: fill.
template <typename T, bool FILL>
__global__ void k_internal_build(const int* __restrict__ lptr, const int* __restrict__ lcol,
                                 const T* __restrict__ lval, int N,
                                 const int* __restrict__ perm, const int* __restrict__ iperm,
                                 int* __restrict__ cnt, int* __restrict__ rptr,
                                 int* __restrict__ rcol, T* __restrict__ rval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // internal row
  if (i >= N) return;
  const int old = perm ? perm[i] : i;
  const int s = lptr[old], e = lptr[old + 1];
  bool has_diag = false;
  for (int j = s; j < e; ++j)
    if (lcol[j] == old) { has_diag = true; break; }
  const int n = (e - s) + (has_diag ? 0 : 1);
  const int npad = (n + 3) & ~3;
  if (!FILL) {
    cnt[i] = npad;
    return;
  }
  const int o = rptr[i];  // multiple of 4 (every row length is)
  rptr[i] = o | (npad - n);  // low 2 bits: number of pad entries closing this row
  // Rows of at most CAP entries (all but hubs): read once into registers, every entry's final position - the
  // diagonal first, the others by ascending internal column - is its rank among the row's entries (CAP^2 predicated
  // compares on registers), written once.  (The insertion sort in global memory below cost 5.2 GB of traffic for
  // 130 MB of matrix at N = 1M: profiles/r04_setup_hostpipe_rocprofv3_summary.txt.)
  constexpr int CAP = sizeof(T) == 8 ? 24 : 32;  // (32 doubles + 32 columns would spill at the default register bound)
  if (n <= CAP) {
    int cc[CAP];
    T vv[CAP];
    const int len = e - s;
#pragma unroll
    for (int p = 0; p < CAP; ++p) {
      const bool in = p < len;
      const int c = in ? lcol[s + p] : 0;
      cc[p] = in ? (iperm ? iperm[c] : c) : (p == len && !has_diag ? i : 0x7FFFFFFF);
      vv[p] = in ? lval[s + p] : T(0);
    }
#pragma unroll
    for (int p = 0; p < CAP; ++p) {
      if (p < n) {
        int pos = 0;
        if (cc[p] != i) {
          pos = 1;
#pragma unroll
          for (int q = 0; q < CAP; ++q) pos += (q < n && cc[q] != i && cc[q] < cc[p]) ? 1 : 0;
        }
        rcol[o + pos] = cc[p];
        rval[o + pos] = vv[p];
      }
    }
    for (int m2 = n; m2 < npad; ++m2) {
      rcol[o + m2] = N;  // out-of-range sentinel: the gather's bounds check returns 0
      rval[o + m2] = T(0);
    }
    return;
  }
  int m = 0;
  for (int j = s; j < e; ++j) {
    rcol[o + m] = iperm ? iperm[lcol[j]] : lcol[j];
    rval[o + m] = lval[j];
    ++m;
  }
  if (!has_diag) {
    rcol[o + m] = i;
    rval[o + m] = T(0);
    ++m;
  }
  // keep short rows sorted by (internal) column: neighbouring gathers stay adjacent
  if ((perm || !has_diag) && m <= 128) {
    for (int p = 1; p < m; ++p) {
      const int c = rcol[o + p];
      const T v = rval[o + p];
      int q = p - 1;
      while (q >= 0 && rcol[o + q] > c) {
        rcol[o + q + 1] = rcol[o + q];
        rval[o + q + 1] = rval[o + q];
        --q;
      }
      rcol[o + q + 1] = c;
      rval[o + q + 1] = v;
    }
  }
  // the diagonal slot becomes entry 0 of the row: the step kernels take T_{k-1}[row] from that
  // gather instead of loading it again (flush and Newton-form steps)
  {
    int p = 0;
    while (p < m && rcol[o + p] != i) ++p;
    const T dv = rval[o + p];
    if (m <= 128) {
      for (int q = p; q > 0; --q) {  // keep the rest sorted
        rcol[o + q] = rcol[o + q - 1];
        rval[o + q] = rval[o + q - 1];
      }
    } else {
      rcol[o + p] = rcol[o];
      rval[o + p] = rval[o];
    }
    rcol[o] = i;
    rval[o] = dv;
  }
  for (; m < npad; ++m) {
    rcol[o + m] = N;  // out-of-range sentinel: the gather's bounds check returns 0
    rval[o + m] = T(0);
  }
}

// F = (2/a1) * (L - a2 I) on the internal layout (approximations.py:105)
template <typename T>
__global__ void k_factor(const int* __restrict__ rptr, const int* __restrict__ rcol,
                         const T* __restrict__ rval, int N, T two_over_a1, T a2,
                         T* __restrict__ fval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int j = rptr[i] & ~3; j < (rptr[i + 1] & ~3); ++j) {
    const int c = rcol[j];
    T v = rval[j];
    if (c == i) v -= a2;
    fval[j] = (c == N) ? T(0) : two_over_a1 * v;
  }
}

// ---------------------------------------------------------------------------------------------
// small vector kernels for the device Lanczos estimate of lambda_max (graph.py:907-920)
// ---------------------------------------------------------------------------------------------
// partial[b] = sum over the block's grid-stride elements of x*y (double accumulation)
template <typename T>
__global__ __launch_bounds__(256) void k_dot_partial(const T* __restrict__ x, const T* __restrict__ y,
                                                     size_t n, double* __restrict__ partial) {
  __shared__ double ws[4];
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += (double)x[i] * (double)y[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// out[0] = sum(partial[0..n)) in a fixed order (deterministic)
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int n,
                                                      double* __restrict__ out) {
  __shared__ double ws[4];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ws[0] + ws[1] + ws[2] + ws[3];
}
// y = a*x + b*y
template <typename T>
__global__ void k_axpby(T a, const T* __restrict__ x, T b, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}
// deterministic start vector: a fixed hash of the index mapped to [-1, 1)
template <typename T> __global__ void k_start_vector(T* v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    v[i] = (T)((double)h / 2147483648.0 - 1.0);
  }
}

// calibration: every workgroup streams the same `n4` float4s `passes` times (read-only), so the
// data is served by whichever cache level holds `16*n4` bytes
__global__ __launch_bounds__(256) void k_read_loop(const float4* __restrict__ p, size_t n4, int passes,
                                                   float* __restrict__ sink) {
  float acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (int r = 0; r < passes; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = p[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 123456.789f) sink[0] = acc;  // keep the loads alive
}

template <typename T> __global__ void k_fill(T* p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of int32 (three small kernels; tile = 1024 elements)
// ---------------------------------------------------------------------------------------------
#define GSPX_SCAN_TILE 1024
// (in and out may be the same array - radix_argsort scans its histogram in place: every thread reads its four
// inputs before it writes its four outputs, and no thread touches another's - so neither is __restrict__)
__global__ __launch_bounds__(256) void k_scan_tiles(const int* in, int n, int* out, int* __restrict__ tile_sums) {
  __shared__ int wsum[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int base = blockIdx.x * GSPX_SCAN_TILE + t * 4;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (base + k < n) ? in[base + k] : 0;
  const int mine = v[0] + v[1] + v[2] + v[3];
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int wbase = 0;
  for (int k = 0; k < wv; ++k) wbase += wsum[k];
  int run = wbase + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (t == 255) tile_sums[blockIdx.x] = wbase + incl;
}

__global__ __launch_bounds__(256) void k_scan_sums(int* tile_sums, int ntiles) {
  // single workgroup: serial over 256-element strips with a carry
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 256) {
    const int i = base + t;
    const int mine = i < ntiles ? tile_sums[i] : 0;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int wbase = carry_s;
    for (int k = 0; k < wv; ++k) wbase += wsum[k];
    if (i < ntiles) tile_sums[i] = wbase + incl - mine;
    __syncthreads();
    if (t == 255) carry_s = wbase + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_scan_add(int* __restrict__ out, int n,
                                                  const int* __restrict__ tile_sums) {
  const int add = tile_sums[blockIdx.x];
  const int base = blockIdx.x * GSPX_SCAN_TILE + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) out[base + k] += add;
}

}  // namespace gspx
