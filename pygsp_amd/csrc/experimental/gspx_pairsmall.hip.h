// gspx_pairsmall.hip.h - two recurrence orders per launch for the LATENCY case: one or two signals on a graph whose
// matrix and panels stay in the L2s (BASELINE config 1: Sensor(100000), one fp64 signal, order 30).  Included by
// gspx.hip in front of run_batch.
//
// Such a call is 30 dependent launches of ~2 us of work each; what it costs is the launch boundary (a global
// synchronisation: every row needs its neighbours' T_{k-1}), ~5 us per order even replayed as one hipGraph
// (DESIGN 7.2).  Halving the number of boundaries is the only lever.  One workgroup per 64-row block of the gather
// tiles (gspx_tile_kernels.hip.h: the block's 1-hop closure S1 and the 8-bit positions of its entries in it):
//   phase 1   A = sA F P + gA Q  on the S1 rows, gathered from global memory (cache resident: the ~2x redundant halo
//             rows cost nothing), kept in LDS; own rows also stored
//   phase 2   B = sB F A + gB P  on the block's rows from LDS, and the flush of up to four terms
// (approximations.py:107-109, twice).  No LDS-DMA, no persistent loop, one barrier: an ordinary kernel.  Same
// polynomial as the single-step path; the final sum is associated differently (four terms per flush).
#pragma once

namespace gspx {

template <typename T> struct PairSmallArgs {
  const int* rowptr;  // internal padded CSR
  const int* col;
  const T* val;       // factor values F
  const int* hdr;     // gather tiles: [nb][4] s1lo, n1, rp0, ent
  const int* s1rows;
  const u8* lidx;
  const T *P, *Q;
  T *A, *B, *R, *y;
  const int* perm;
  int N;
  unsigned ldy;
  T sA, gA, sB, gB;
  T wB, wA, wP, wQ;
  int flush;  // 0 none, 1 write R, 2 accumulate into R
  int final;  // 1: the flush result goes to y (caller's vertex order)
};

// LD signals per row (1 or 2: a row is at most 16 bytes).  512 threads = 128 groups of 4 lanes; a group takes a row,
// its lanes every fourth entry (rows are padded to multiples of four), the partial sums meet in two shuffles.
// (First version: one thread per row - 18 us per pair launch against 2 x 5.1 us of single steps: a thread's
// column -> panel-row load chain, twelve times in a row, is pure latency.)
template <typename T, int LD>
__global__ __launch_bounds__(512) void k_pair_small(const PairSmallArgs<T> a) {
  __shared__ T tile_g[GSPX_TILE_MAXN1 * LD];
  const int b = (int)blockIdx.x;
  const int4 h = *(const int4*)(a.hdr + (size_t)b * 4);
  const int n1 = h.y;  // (>= 0: the host takes this path only when every block has its list)
  const int r0 = b * GSPX_TILE_BR;
  const int lane = (int)threadIdx.x & 3, grp = (int)threadIdx.x >> 2;
  const int nlast = a.N - 1;
  for (int o = grp; o < n1; o += 128) {
    const int r = a.s1rows[h.x + o];
    const int s = a.rowptr[r] & ~3, e = a.rowptr[r + 1] & ~3;
    T acc[LD];
#pragma unroll
    for (int l = 0; l < LD; ++l) acc[l] = T(0);
#pragma unroll 4
    for (int j = s + lane; j < e; j += 4) {
      const int c = a.col[j];
      const T v = a.val[j];
      const int cc = c < nlast ? c : nlast;  // (pads carry column N and value 0: their product is dropped)
#pragma unroll
      for (int l = 0; l < LD; ++l) {
        const T pv = a.P[(size_t)cc * LD + l];
        acc[l] += c <= nlast ? v * pv : T(0);
      }
    }
#pragma unroll
    for (int l = 0; l < LD; ++l) {
      acc[l] += __shfl_xor(acc[l], 1);
      acc[l] += __shfl_xor(acc[l], 2);
    }
    if (lane == 0) {
      const bool own = r >= r0 && r < r0 + GSPX_TILE_BR;
#pragma unroll
      for (int l = 0; l < LD; ++l) {
        T av = a.sA * acc[l];
        if (a.gA != T(0)) av += a.gA * a.Q[(size_t)r * LD + l];
        tile_g[o * LD + l] = av;
        if (own) a.A[(size_t)r * LD + l] = av;
      }
    }
  }
  __syncthreads();
  const int row = r0 + grp;
  if (grp < GSPX_TILE_BR && row < a.N) {
    const int s = a.rowptr[row] & ~3, e = a.rowptr[row + 1] & ~3;
    T acc[LD];
#pragma unroll
    for (int l = 0; l < LD; ++l) acc[l] = T(0);
#pragma unroll 4
    for (int j = s + lane; j < e; j += 4) {
      const int p = (int)a.lidx[j];
      const T v = a.val[j];
#pragma unroll
      for (int l = 0; l < LD; ++l) acc[l] += v * tile_g[p * LD + l];
    }
#pragma unroll
    for (int l = 0; l < LD; ++l) {
      acc[l] += __shfl_xor(acc[l], 1);
      acc[l] += __shfl_xor(acc[l], 2);
    }
    if (lane == 0) {
      const size_t orow = a.perm ? (size_t)a.perm[row] : (size_t)row;
#pragma unroll
      for (int l = 0; l < LD; ++l) {
        const T self = tile_g[(int)a.lidx[s] * LD + l];  // entry 0 of a row is its diagonal slot: A[row]
        const T p = a.P[(size_t)row * LD + l];
        const T bv = a.sB * acc[l] + a.gB * p;
        a.B[(size_t)row * LD + l] = bv;
        if (a.flush) {
          T res = a.wB * bv + a.wA * self + a.wP * p;
          if (a.wQ != T(0)) res += a.wQ * a.Q[(size_t)row * LD + l];
          if (a.flush == 2) res += a.R[(size_t)row * LD + l];
          if (a.final) a.y[orow * a.ldy + l] = res;
          else a.R[(size_t)row * LD + l] = res;
        }
      }
    }
  }
}

}  // namespace gspx

// the call shapes this path takes (see run_batch): one filter of even order, rows of at most 16 bytes, every block of
// the gather tiles with its list, matrix + panels small enough to stay in the caches
template <typename T>
static bool pair_small_usable(const gspx_graph* g, const Options& opt, int nf, int M, unsigned ld, bool deferred,
                              bool acc_existing, bool final_to_y) {
  const int K = M - 1;
  return opt.pair_small && nf == 1 && !deferred && !acc_existing && final_to_y && K >= 2 && (K & 1) == 0 &&
         (ld == 1 || (ld == 2 && sizeof(T) * 2 <= 16)) && g->gt_rows == GSPX_TILE_BR && g->gt_slow == 0 && g->gt_nb > 0 &&
         (size_t)g->nnz_int * (sizeof(T) + 4) < ((size_t)opt.pair_small_mb << 20);
}

template <typename T>
static int run_pair_small(gspx_graph* g, int M, const std::vector<double>& c, const T* x, unsigned ldx, T* y,
                          unsigned ldy, unsigned ld, size_t& ev_idx) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int N = (int)g->N, K = M - 1;
  const size_t U = (size_t)N * ld;
  const bool cap = ctx->capturing;  // (recording a replay: the eager call before left the workspaces in place)
  CHK(ctx->ws_t.ensure(4 * U * sizeof(T) + 256));
  CHK(ctx->ws_r.ensure(U * sizeof(T) + 256));
  T* X[4] = {ctx->ws_t.as<T>(), ctx->ws_t.as<T>() + U, ctx->ws_t.as<T>() + 2 * U, ctx->ws_t.as<T>() + 3 * U};
  const int* perm = g->has_perm ? g->perm.as<int>() : nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
  if (!cap) {
    e0 = pool_event(ctx, ++ev_idx);
    e1 = pool_event(ctx, ++ev_idx);
    e2 = pool_event(ctx, ++ev_idx);
    e3 = pool_event(ctx, ++ev_idx);
    if (!e0 || !e1 || !e2 || !e3) return set_err(GSPX_ERR_HIP, "hipEventCreate failed");
    HIPCHK(hipEventRecord(e0, st));
  }
  launch_permute_in<T>(x, ldx, X[0], ld, N, perm, 1, st);
  if (!cap) HIPCHK(hipEventRecord(e1, st));

  gspx::PairSmallArgs<T> a{};
  a.rowptr = g->rptr.as<int>();
  a.col = g->rcol.as<int>();
  a.val = g->fval.as<T>();
  a.hdr = g->gt_hdr.as<int>();
  a.s1rows = g->gt_s1rows.as<int>();
  a.lidx = g->gt_lidx.as<unsigned char>();
  a.perm = perm;
  a.N = N;
  a.y = y;
  a.ldy = ldy;
  a.R = ctx->ws_r.as<T>();
  // pair j = orders 2j - 1 and 2j:  P = T_{2j-2}, Q = T_{2j-3};  A = T_{2j-1} and B = T_{2j} go to the two free panels
  int iP = 0, iQ = -1, iA = 1, iB = 2, spare = 3;
  const int npairs = K / 2;
  int folded = 0;  // highest order in the running sum so far
  for (int j = 1; j <= npairs; ++j) {
    a.P = X[iP];
    a.Q = iQ >= 0 ? X[iQ] : X[iP];  // (never read for its value when gA = wQ = 0)
    a.A = X[iA];
    a.B = X[iB];
    if (j == 1) {  // T_1 = (L - a2) x / a1 = F x / 2;  T_2 = F T_1 - T_0
      a.sA = T(0.5); a.gA = T(0); a.sB = T(1); a.gB = T(-1);
    } else {
      a.sA = T(1); a.gA = T(-1); a.sB = T(1); a.gB = T(-1);
    }
    const bool last = j == npairs;
    const bool flush = j == 1 || last || (j & 1) == 1;
    a.flush = a.final = 0;
    a.wB = a.wA = a.wP = a.wQ = T(0);
    if (flush) {
      a.flush = j == 1 ? 1 : 2;
      a.final = last ? 1 : 0;
      a.wB = (T)c[(size_t)2 * j];
      a.wA = (T)c[(size_t)2 * j - 1];
      if (j == 1) a.wP = (T)c[0];  // (make_plan's convention: c[0] arrives already halved)
      else if (folded < 2 * j - 2) {
        a.wP = (T)c[(size_t)2 * j - 2];
        a.wQ = (T)c[(size_t)2 * j - 3];
      }
      folded = 2 * j;
    }
    if (ld == 1) hipLaunchKernelGGL((gspx::k_pair_small<T, 1>), dim3((unsigned)g->gt_nb), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gspx::k_pair_small<T, 2>), dim3((unsigned)g->gt_nb), dim3(512), 0, st, a);
    const int oP = iP, oQ = iQ >= 0 ? iQ : spare;
    iP = iB;
    iQ = iA;
    iA = oQ;
    iB = oP;
  }
  if (!cap) {
    HIPCHK(hipEventRecord(e2, st));
    HIPCHK(hipEventRecord(e3, st));
    HIPCHK(hipGetLastError());
  }
  return GSPX_OK;
}
