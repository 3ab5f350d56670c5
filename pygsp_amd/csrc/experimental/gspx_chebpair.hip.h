// gspx_chebpair.hip.h - TWO orders of the three-term Chebyshev recurrence per launch (round-4 experiment,
// opt-in: gspx_graph_set_cheb_pair_tiles + gspx_cheby_pair_filter_dev).
//
//   A = sA F P + gA Q      on the block's 1-hop closure S1     (T_k   = 2/a1 (L - a2) T_{k-1} - T_{k-2})
//   B = sB F A + gB P      on the block's own rows             (T_k+1 = 2/a1 (L - a2) T_k     - T_{k-1})
//   approximations.py:107-109, twice; the flush folds up to four terms c_k T_k at hand into the running sum.
//
// Why: a single step moves CSR + 3U + 2/3 U of accumulator per order and runs at the memory system's rate
// (DESIGN 7.1); the only byte lever left inside the recurrence is temporal blocking.  A pair launch reads P on the
// 2-hop closure S2 (staged in LDS by LDS-DMA, its halo served by the XCD's L2), Q on S1, writes A and B on the own
// rows: ~4.2U + one accumulator read-modify-write every second launch, i.e. ~2.6U per order instead of 3.67U -
// paid for with (|S1| + |B|) / (2 |B|) row products per output row (1.3 at 128-row blocks) and 2.3 L2-level reads of
// P per row.  Blocks of BR rows (64 / 128 / 256), column chunks of CW lanes x 16 bytes so that both tiles and the
// S1 rows' matrix entries fit a workgroup's LDS.
//
// Tile data (pygsp_amd/tiling.py, levels = 2): per block the ascending row lists S1 and S2, for every entry of an
// S1 row its position in S2 (lidx2, laid out per (block, S1 row) occurrence: contiguous per block, so are the
// gathered values val2), for every entry of an own row its position in S1 (lidx1), for every row its own
// position in its block's S1 (ownpos).
#pragma once

namespace gspx {

template <typename T> struct ChebPairArgs {
  const int* rowptr;     // internal padded CSR row starts (low 2 bits: pad counts)
  const int* hdr;        // [nb][8]: s1lo, n1, s2lo, n2, occ_base, ent1, rp0, ent2
  const int2* desc;      // [n_s1]: row, local entry offset | padded length << 24
  const int* s2rows;
  const T* val2;         // [total2] factor values per (block, S1 row) occurrence
  const u16* lidx2;      // [total2] positions in S2
  const u16* lidx1;      // [nnz_int] positions in S1 (own rows)
  const u16* ownpos;     // [N]
  const int2* owndesc;   // [N]: the row's entry offset | padded length << 24 in its block's slice, offset of its S1 positions
  const T *P, *Q;
  T *A, *B, *R, *y;
  const int* perm;
  int N, BR, nb, per_xcd, ncol;
  u32 ld, ldy, panel_bytes;
  int off_g, off_val, off_idx1, off_idx2;  // LDS byte offsets (tile_h at 0)
  int off_ms2, off_ms1, off_mown;          // k_cheb_pair2: the block's row lists / descriptors in LDS (first copy)
  int meta_bytes;                          // ... and the distance to the second copy
  u32 s2rows_bytes, desc_bytes, owndesc_bytes;
  u32 val2_bytes, lidx2_bytes, lidx1_bytes;
  unsigned long long* dbg;                 // GSPX_PAIR_DEBUG=1: per workgroup 8 cycle sums (wave 0), else null
  T sA, gA, sB, gB;
  T wB, wA, wP, wQ;
  int flush;   // 0 none, 1 write R, 2 accumulate into R
  int final;   // 1: the flush result goes to y (caller's vertex order)
};

// CW lanes (16-byte pieces) per row and column chunk; 512 threads = NG = 512 / CW row groups
template <typename T, int CW>
__global__ __launch_bounds__(512, 4) void k_cheb_pair(const ChebPairArgs<T> a) {
  constexpr int NT = 512, NG = NT / CW, RB = CW * 16;
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef typename VT<T, VEC>::t V;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef unsigned long long u64;
  extern __shared__ __attribute__((aligned(16))) unsigned char gspx_smem[];
  constexpr u32 POISON = 0x80000000u;

  const int tid = threadIdx.x, lane = tid % CW, grp = tid / CW, wave = tid >> 6;
  const int nwx = (int)(gridDim.x >> 3);
  const int xlo = (int)(blockIdx.x & 7) * a.per_xcd;
  int k1 = xlo + a.per_xcd;
  if (k1 > a.nb) k1 = a.nb;
  int k = xlo + (int)(blockIdx.x >> 3);
  if (k >= k1) return;

  const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.P, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)a.Q, 0, a.Q ? a.panel_bytes : 0, 0x00020000);
  const rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)a.R, 0, a.R ? a.panel_bytes : 0, 0x00020000);
  const u32 ldb = a.ld * (u32)sizeof(T);

  V* const tile_h = (V*)gspx_smem;
  V* const tile_g = (V*)(gspx_smem + a.off_g);
  T* const val1 = (T*)(gspx_smem + a.off_val);
  u16* const idx1 = (u16*)(gspx_smem + a.off_idx1);
  u16* const idx2 = (u16*)(gspx_smem + a.off_idx2);

  auto load_hdr = [&](int kk, int* h) {
    const int4 p = *(const int4*)(a.hdr + (size_t)kk * 8), q = *(const int4*)(a.hdr + (size_t)kk * 8 + 4);
    h[0] = __builtin_amdgcn_readfirstlane(p.x); h[1] = __builtin_amdgcn_readfirstlane(p.y);
    h[2] = __builtin_amdgcn_readfirstlane(p.z); h[3] = __builtin_amdgcn_readfirstlane(p.w);
    h[4] = __builtin_amdgcn_readfirstlane(q.x); h[5] = __builtin_amdgcn_readfirstlane(q.y);
    h[6] = __builtin_amdgcn_readfirstlane(q.z); h[7] = __builtin_amdgcn_readfirstlane(q.w);
  };
  auto chunk_off = [&](int c) {
    const u32 col0 = (c * CW + lane) * VEC;
    return col0 < a.ld ? col0 * (u32)sizeof(T) : POISON;
  };
  // P on the S2 rows, column chunk c -> tile_h, straight into LDS: a wave instruction writes 64 / CW tile rows
  auto stage = [&](const int* h, int c) {
    const u32 cb = chunk_off(c);
    const int n2 = h[3];
    for (int t = 0; grp - (grp % (64 / CW)) + NG * t < n2; ++t) {  // (wave-uniform trip count)
      const int u = grp + NG * t;
      const int row = a.s2rows[h[2] + (u < n2 ? u : n2 - 1)];
      if (u < n2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rP, (lds_ptr)(gspx_smem + (wave * (64 / CW) + NG * t) * RB), 16,
                                                 (u32)row * ldb + cb, 0, 0, 0);
    }
  };

  int H[8];
  load_hdr(k, H);
  stage(H, 0);
  for (;;) {
    const int s1lo = H[0], n1 = H[1], occ_base = H[4], ent1 = H[5], rp0 = H[6], ent2 = H[7];
    const int r0 = k * a.BR;
    // the block's matrix entries -> LDS, once for all its column chunks (coalesced slices)
    {
      const int nv16 = (ent1 * (int)sizeof(T) + 15) >> 4;
      const u32x4* src = (const u32x4*)(a.val2 + occ_base);
      for (int i = tid; i < nv16; i += NT) ((u32x4*)val1)[i] = src[i];
      const u64* s1 = (const u64*)(a.lidx2 + occ_base);
      for (int i = tid; i < (ent1 >> 2); i += NT) ((u64*)idx1)[i] = s1[i];
      const u64* s2 = (const u64*)(a.lidx1 + rp0);
      for (int i = tid; i < (ent2 >> 2); i += NT) ((u64*)idx2)[i] = s2[i];
    }
    int kn = k + nwx;
    int Hn[8];
    load_hdr(kn < k1 ? kn : k, Hn);
    for (int c = 0; c < a.ncol; ++c) {
      const u32 cb = chunk_off(c);
      const bool on = cb != POISON;
      const u32 col0 = (c * CW + lane) * VEC;
      // every wave waits for its own tile loads (`buffer_load ... lds` is tracked by vmcnt only)
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
      __syncthreads();                     // tile_h and the entries in place; tile_g free
      // ---- phase 1: A on S1 ---------------------------------------------------------------------------
      for (int o = grp; o < n1; o += NG) {
        const int2 d = a.desc[s1lo + o];
        const int row = d.x, off = d.y & 0xFFFFFF, len = (int)((unsigned)d.y >> 24);
        V q = 0;
        if (a.gA != T(0)) q = VT<T, VEC>::bload(rQ, on ? (u32)row * ldb + cb : POISON);
        V self;
        const V acc = lds_row_dot<T, V, CW, u16>(val1 + off, idx1 + off, len, tile_h, lane, self);
        const V av = a.sA * acc + a.gA * q;
        tile_g[o * CW + lane] = av;
        if (on && row >= r0 && row < r0 + a.BR) *(V*)(a.A + (size_t)row * a.ld + col0) = av;
      }
      __syncthreads();  // tile_g complete; everybody is done with tile_h
      // the next pass's tile: the next chunk of this block, or chunk 0 of the next block
      const bool last = c == a.ncol - 1;
      if (!last) stage(H, c + 1);
      else if (kn < k1) stage(Hn, 0);
      // ---- phase 2: B on the own rows -----------------------------------------------------------------
      for (int rr = grp; rr < a.BR; rr += NG) {
        const int row = r0 + rr;
        if (row >= a.N) break;
        const int o = a.ownpos[row];
        const int2 d = a.desc[s1lo + o];
        const int off = d.y & 0xFFFFFF, len = (int)((unsigned)d.y >> 24);
        const int loc = (a.rowptr[row] & ~3) - rp0;
        const u32 ro = on ? (u32)row * ldb + cb : POISON;
        const V p = VT<T, VEC>::bload(rP, ro);
        V qq = 0, ra = 0;
        if (a.flush && a.wQ != T(0)) qq = VT<T, VEC>::bload(rQ, ro);
        if (a.flush == 2) ra = VT<T, VEC>::bload(rR, ro);
        V self;
        const V acc = lds_row_dot<T, V, CW, u16>(val1 + off, idx2 + loc, len, tile_g, lane, self);
        const V bv = a.sB * acc + a.gB * p;
        if (on) {
          *(V*)(a.B + (size_t)row * a.ld + col0) = bv;
          if (a.flush) {
            V res = a.wB * bv + a.wA * self + a.wP * p + a.wQ * qq;
            if (a.flush == 2) res += ra;
            if (a.final) {
              const size_t orow = a.perm ? (size_t)a.perm[row] : (size_t)row;
              *(V*)(a.y + orow * a.ldy + col0) = res;
            } else {
              *(V*)(a.R + (size_t)row * a.ld + col0) = res;
            }
          }
        }
      }
    }
    if (kn >= k1) break;
    __syncthreads();  // everybody is done with this block's entries before the next block's arrive
    k = kn;
#pragma unroll
    for (int i = 0; i < 8; ++i) H[i] = Hn[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_cheb_pair2 (round 5): the same two orders per launch with the memory round trips taken off the critical path.
// Round 5 measured the kernel above at size for the first time (tools/pair_ladder.py: parity-green at N = 1M, 0.40 of
// 8 TB/s against 0.62 for single steps): every pass waited for chains of dependent global loads - the S2 row list
// before each LDS-DMA instruction, the S1 descriptors and then the T_{k-2} rows inside phase 1, ownpos -> descriptor ->
// rowptr -> rows inside phase 2, the block's entries at its first pass.  Here
//   * a block's row lists / descriptors / own-row records live in LDS (two copies: the next block's are written during
//     the current block's last pass), its matrix entries next to them: all loaded ONCE per block (its column chunks
//     reuse them), fetched into registers one pass ahead;
//   * the next tile's LDS-DMA and the T_{k-2} rows of the next phase 1 are issued right after the second barrier -
//     the tile of the NEXT BLOCK's first chunk too - so they are in flight during phase 2; the own rows' T_{k-1} /
//     T_{k-2} / accumulator values follow after the stores;
//   * the LDS-DMA instructions of a tile are issued back to back (row numbers come from LDS);
//   * row products fetch the next four entries while the tile rows of the current four are in flight.
// A per-pass cycle breakdown (GSPX_PAIR_DEBUG=1, DBG builds) guided this: profiles/r05_pair_experiment.md.
// IT1 / IT2: rows of S1 / of the block a row group handles per pass (IT1 * NG >= largest S1, IT2 * NG >= block rows).
template <typename T, typename V, int LG, typename I>
__device__ __forceinline__ V lds_row_dot_pf(const T* val, const I* idx, int len, const V* tile, int lane, V& self) {
  typedef T T4 __attribute__((ext_vector_type(4)));
  typedef I I4 __attribute__((ext_vector_type(4)));
  V acc = 0;
  self = 0;
  if (len <= 0) return acc;
  I4 ia = *(const I4*)idx;
  T4 va = *(const T4*)val;
  for (int j = 0; j < len; j += 4) {
    const int jn = j + 4 < len ? j + 4 : j;
    const I4 ib = *(const I4*)(idx + jn);  // the next four entries: in flight with this trip's tile rows
    const T4 vb = *(const T4*)(val + jn);
    const V t0 = tile[ia.x * LG + lane];
    const V t1 = tile[ia.y * LG + lane];
    const V t2 = tile[ia.z * LG + lane];
    const V t3 = tile[ia.w * LG + lane];
    if (j == 0) self = t0;
    acc += va.x * t0;
    acc += va.y * t1;
    acc += va.z * t2;
    acc += va.w * t3;
    ia = ib;
    va = vb;
  }
  return acc;
}

template <typename T, int CW, int IT1, int IT2, bool DBG = false>
__global__ __launch_bounds__(512, 4) void k_cheb_pair2(const ChebPairArgs<T> a) {
  typedef unsigned long long u64;
  u64 dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // DBG: cycles of wave 0 in: tile wait, barrier A, phase 1, barrier B, phase 2, issue, passes, block change
  constexpr int NT = 512, NG = NT / CW, RB = CW * 16, WR = 64 / CW;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int NEV = sizeof(T) == 8 ? 3 : 2;  // 16-byte pieces of matrix values a thread carries across a block boundary
  typedef typename VT<T, VEC>::t V;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) unsigned char gspx_smem[];
  constexpr u32 POISON = 0x80000000u;

  const int tid = threadIdx.x, lane = tid % CW, grp = tid / CW, wave = tid >> 6;
  const int nwx = (int)(gridDim.x >> 3);
  const int xlo = (int)(blockIdx.x & 7) * a.per_xcd;
  int k1 = xlo + a.per_xcd;
  if (k1 > a.nb) k1 = a.nb;
  int k = xlo + (int)(blockIdx.x >> 3);
  if (k >= k1) return;

  const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.P, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)a.Q, 0, a.Q ? a.panel_bytes : 0, 0x00020000);
  const rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)a.R, 0, a.R ? a.panel_bytes : 0, 0x00020000);
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)a.val2, 0, a.val2_bytes, 0x00020000);
  const rsrc_t rI2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.lidx2, 0, a.lidx2_bytes, 0x00020000);
  const rsrc_t rI1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.lidx1, 0, a.lidx1_bytes, 0x00020000);
  const u32 ldb = a.ld * (u32)sizeof(T);

  V* const tile_h = (V*)gspx_smem;
  V* const tile_g = (V*)(gspx_smem + a.off_g);
  T* const val1 = (T*)(gspx_smem + a.off_val);
  u16* const idx1 = (u16*)(gspx_smem + a.off_idx1);
  u16* const idx2 = (u16*)(gspx_smem + a.off_idx2);
  // the lists of a block, two copies mb = 0 / 1 (a.meta_bytes apart): rows of S2; per S1 row its row and entry
  // offset | padded length << 24; per own row the same descriptor word and the offset of its entries' S1 positions
  auto ms2 = [&](int mb) { return (int*)(gspx_smem + a.off_ms2 + mb * a.meta_bytes); };
  auto ms1 = [&](int mb) { return (int2*)(gspx_smem + a.off_ms1 + mb * a.meta_bytes); };
  auto mown = [&](int mb) { return (int2*)(gspx_smem + a.off_mown + mb * a.meta_bytes); };

  struct Hdr { int4 p, q; };  // p: s1lo, n1, s2lo, n2; q: occ_base, ent1, rp0, ent2
  auto load_hdr = [&](int kk) {
    Hdr h;
    h.p = *(const int4*)(a.hdr + (size_t)kk * 8);
    h.q = *(const int4*)(a.hdr + (size_t)kk * 8 + 4);
    return h;
  };
  auto uniform = [](Hdr h) {
    Hdr u;
    u.p.x = __builtin_amdgcn_readfirstlane(h.p.x); u.p.y = __builtin_amdgcn_readfirstlane(h.p.y);
    u.p.z = __builtin_amdgcn_readfirstlane(h.p.z); u.p.w = __builtin_amdgcn_readfirstlane(h.p.w);
    u.q.x = __builtin_amdgcn_readfirstlane(h.q.x); u.q.y = __builtin_amdgcn_readfirstlane(h.q.y);
    u.q.z = __builtin_amdgcn_readfirstlane(h.q.z); u.q.w = __builtin_amdgcn_readfirstlane(h.q.w);
    return u;
  };
  const rsrc_t rS2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.s2rows, 0, a.s2rows_bytes, 0x00020000);
  const rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)a.desc, 0, a.desc_bytes, 0x00020000);
  const rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)a.owndesc, 0, a.owndesc_bytes, 0x00020000);
  // A contiguous slice of global memory -> LDS by LDS-DMA, no registers: n pieces of PB (4 or 16) bytes; a wave
  // instruction moves 64 consecutive pieces.  Completion is tracked by vmcnt like the tile loads.
  auto dma16 = [&](rsrc_t r, u32 src, int n, unsigned char* dst) {
    for (int i0 = wave * 64; i0 < n; i0 += NT) {  // (wave-uniform)
      const int i = i0 + (tid & 63);
      if (i < n) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + i0 * 16), 16, src + (u32)i * 16u, 0, 0, 0);
    }
  };
  auto dma4 = [&](rsrc_t r, u32 src, int n, unsigned char* dst) {
    for (int i0 = wave * 64; i0 < n; i0 += NT) {
      const int i = i0 + (tid & 63);
      if (i < n) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + i0 * 4), 4, src + (u32)i * 4u, 0, 0, 0);
    }
  };
  // a block's lists -> list copy mb (rows of S2, S1 descriptors, own-row records) ...
  auto stage_meta = [&](const Hdr& h, int kk, int mb) {
    dma4(rS2, (u32)h.p.z * 4u, h.p.w, (unsigned char*)ms2(mb));
    dma4(rD, (u32)h.p.x * 8u, h.p.y * 2, (unsigned char*)ms1(mb));
    int nown = a.N - kk * a.BR;
    if (nown > a.BR) nown = a.BR;
    dma4(rO, (u32)kk * (u32)a.BR * 8u, nown * 2, (unsigned char*)mown(mb));
  };
  // ... and its matrix entries (values, positions in S2 of the S1 rows' entries, positions in S1 of the own rows')
  auto stage_entries = [&](const Hdr& h) {
    const int ent1 = h.q.y, rp0 = h.q.z, ent2 = h.q.w;
    const u32 occ = (u32)h.q.x;
    dma16(rV, occ * (u32)sizeof(T), (ent1 * (int)sizeof(T) + 15) >> 4, (unsigned char*)val1);
    dma16(rI2, occ * 2u, (ent1 * 2 + 15) >> 4, (unsigned char*)idx1);
    dma16(rI1, (u32)rp0 * 2u, (ent2 * 2 + 15) >> 4, (unsigned char*)idx2);
  };
  auto chunk_off = [&](int c) {
    const u32 col0 = (c * CW + lane) * VEC;
    return col0 < a.ld ? col0 * (u32)sizeof(T) : POISON;
  };
  // T_{k-1} on the S2 rows (list copy mb), column chunk at cb -> tile_h by LDS-DMA: a wave instruction writes WR tile rows
  auto stage = [&](int mb, int n2, u32 cb) {
    const int* rows = ms2(mb);
    for (int t = 0; grp - (grp % WR) + NG * t < n2; ++t) {  // (wave-uniform trip count)
      const int u = grp + NG * t;
      if (u < n2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rP, (lds_ptr)(gspx_smem + (wave * WR + NG * t) * RB), 16,
                                                 (u32)rows[u] * ldb + cb, 0, 0, 0);
    }
  };
  // the rows a pass reads from global memory, one pass ahead: T_{k-2} on the group's S1 rows (phase 1) ...
  V q1[IT1], pp[IT2], qq[IT2], ra[IT2];
  auto prefetch_q1 = [&](int mb, int n1, u32 cb) {
    const int2* d1 = ms1(mb);
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
      const int o = grp + NG * it;
      u32 off = POISON;
      if (o < n1 && cb != POISON && a.gA != T(0)) off = (u32)d1[o].x * ldb + cb;
      q1[it] = VT<T, VEC>::bload(rQ, off);
    }
  };
  // ... and T_{k-1} / T_{k-2} / the accumulator on its own rows (phase 2)
  auto prefetch_own = [&](int r0, u32 cb) {
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
      const int rr = grp + NG * it, row = r0 + rr;
      const u32 off = (rr < a.BR && row < a.N && cb != POISON) ? (u32)row * ldb + cb : POISON;
      pp[it] = VT<T, VEC>::bload(rP, off);
      qq[it] = VT<T, VEC>::bload(rQ, (a.flush && a.wQ != T(0)) ? off : POISON);
      ra[it] = VT<T, VEC>::bload(rR, a.flush == 2 ? off : POISON);
    }
  };

  Hdr H = uniform(load_hdr(k));
  int kn = k + nwx;
  Hdr Hv = load_hdr(kn < k1 ? kn : k);
  int mb = 0;  // the list copy of the current block
  stage_meta(H, k, 0);
  stage_entries(H);
  Hdr Hn = uniform(Hv);
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
  __syncthreads();  // the first block's lists are in LDS
  prefetch_q1(0, H.p.y, chunk_off(0));
  stage(0, H.p.w, chunk_off(0));
  prefetch_own(k * a.BR, chunk_off(0));
  for (;;) {
    const int n1 = H.p.y, n2 = H.p.w;
    const int r0 = k * a.BR;
    const int knn = kn + nwx;
    const bool more = kn < k1;
    for (int c = 0; c < a.ncol; ++c) {
      const u32 cb = chunk_off(c);
      const bool on = cb != POISON;
      const u32 col0 = (c * CW + lane) * VEC;
      const bool last = c == a.ncol - 1;
      // every wave waits for its own tile loads (`buffer_load ... lds` is tracked by vmcnt only) and row prefetches
      const u64 t0 = DBG ? __builtin_readcyclecounter() : 0;
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
      const u64 t1 = DBG ? __builtin_readcyclecounter() : 0;
      __syncthreads();                     // tile_h (and a new block's entries) in place; tile_g free
      const u64 t2 = DBG ? __builtin_readcyclecounter() : 0;
      if (c == 0 && more) stage_meta(Hn, kn, mb ^ 1);  // the next block's lists -> the other copy, read from the last pass on
      if (last) Hv = load_hdr(knn < k1 ? knn : k);     // ... and the header after it
      const int2* const d1 = ms1(mb);
      // ---- phase 1: A on S1 ---------------------------------------------------------------------------
#pragma unroll
      for (int it = 0; it < IT1; ++it) {
        const int o = grp + NG * it;
        if (o < n1) {
          const int2 d = d1[o];
          const int row = d.x, off = d.y & 0xFFFFFF, len = (int)((unsigned)d.y >> 24);
          V self;
          const V acc = lds_row_dot_pf<T, V, CW, u16>(val1 + off, idx1 + off, len, tile_h, lane, self);
          const V av = a.sA * acc + a.gA * q1[it];
          tile_g[o * CW + lane] = av;
          if (on && row >= r0 && row < r0 + a.BR) VT<T, VEC>::bstore(rA, (u32)row * ldb + cb, av);
        }
      }
      const u64 t3 = DBG ? __builtin_readcyclecounter() : 0;
      if (a.ncol == 1) __builtin_amdgcn_s_waitcnt(0x0070);  // (one chunk per block: the lists staged in this very pass)
      __syncthreads();  // tile_g complete; everybody is done with tile_h
      const u64 t4 = DBG ? __builtin_readcyclecounter() : 0;
      // the next pass's T_{k-2} rows and tile - of this block's next chunk, or of the next block's first - go in flight
      // now: phase 2 does not touch tile_h
      if (!last) {
        prefetch_q1(mb, n1, chunk_off(c + 1));
        stage(mb, n2, chunk_off(c + 1));
      } else if (more) {
        prefetch_q1(mb ^ 1, Hn.p.y, chunk_off(0));
        stage(mb ^ 1, Hn.p.w, chunk_off(0));
      }
      const u64 t5 = DBG ? __builtin_readcyclecounter() : 0;
      // ---- phase 2: B on the own rows -----------------------------------------------------------------
#pragma unroll
      for (int it = 0; it < IT2; ++it) {
        const int rr = grp + NG * it, row = r0 + rr;
        if (rr < a.BR && row < a.N) {
          const int2 m = mown(mb)[rr];
          const int off = m.x & 0xFFFFFF, len = (int)((unsigned)m.x >> 24);
          V self;
          const V acc = lds_row_dot_pf<T, V, CW, u16>(val1 + off, idx2 + m.y, len, tile_g, lane, self);
          const V bv = a.sB * acc + a.gB * pp[it];
          if (on) {
            const u32 ro = (u32)row * ldb + cb;
            VT<T, VEC>::bstore(rB, ro, bv);
            if (a.flush) {
              V res = a.wB * bv + a.wA * self + a.wP * pp[it] + a.wQ * qq[it];
              if (a.flush == 2) res += ra[it];
              if (a.final) {
                const size_t orow = a.perm ? (size_t)a.perm[row] : (size_t)row;
                *(V*)(a.y + orow * a.ldy + col0) = res;
              } else {
                VT<T, VEC>::bstore(rR, ro, res);
              }
            }
          }
        }
      }
      // the own rows of the next pass, after this pass's stores
      if (!last) prefetch_own(r0, chunk_off(c + 1));
      else if (more) prefetch_own(kn * a.BR, chunk_off(0));
      if (DBG) {
        const u64 t6 = __builtin_readcyclecounter();
        dsum[0] += t1 - t0; dsum[1] += t2 - t1; dsum[2] += t3 - t2; dsum[3] += t4 - t3; dsum[4] += t6 - t5;
        dsum[5] += t5 - t4; dsum[6] += 1;
      }
    }
    if (!more) break;
    const u64 tb0 = DBG ? __builtin_readcyclecounter() : 0;
    __syncthreads();  // everybody is done with this block's entries
    stage_entries(Hn);  // (complete at the next pass's wait, visible after its first barrier)
    mb ^= 1;
    H = Hn;
    Hn = uniform(Hv);
    k = kn;
    kn = knn;
    if (DBG) dsum[7] += __builtin_readcyclecounter() - tb0;
  }
  if (DBG && a.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = dsum[i];
  }
}

// val2[i] = fval[src[i]]: the factor values per (block, S1 row) occurrence (rebuilt when lmax changes)
template <typename T>
__global__ void k_gather_vals(const T* __restrict__ fval, const int* __restrict__ src, size_t n, T* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fval[src[i]];
}

}  // namespace gspx

extern "C" int gspx_graph_set_cheb_pair_tiles(gspx_graph* g, int block_rows, int nb, const int32_t* s1ptr,
                                              const int32_t* s1rows, const int32_t* s2ptr, const int32_t* s2rows,
                                              const uint16_t* lidx1, const uint32_t* occ_off, int64_t n_lidx2,
                                              const uint16_t* lidx2, int64_t stats[6]) {
  if (g) replay_reset(g->ctx);
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  ChebPairTiles& cp = g->cp;
  if (block_rows == 0) {
    cp.rows = 0;
    return GSPX_OK;
  }
  if (block_rows != 64 && block_rows != 128 && block_rows != 256)
    return set_err(GSPX_ERR_INVALID, "pair tiles use 64-, 128- or 256-row blocks");
  if (!s1ptr || !s1rows || !s2ptr || !s2rows || !lidx1 || !occ_off || !lidx2 || nb < 1 ||
      nb != (int)((g->N + block_rows - 1) / block_rows))
    return set_err(GSPX_ERR_INVALID, "gspx_graph_set_cheb_pair_tiles: bad argument");
  HIPCHK(hipSetDevice(g->ctx->device));
  const int N = (int)g->N;
  const int n_s1 = s1ptr[nb], n_s2 = s2ptr[nb];
  std::vector<int> rp((size_t)N + 1);
  HIPCHK(hipMemcpy(rp.data(), g->rptr.p, ((size_t)N + 1) * sizeof(int), hipMemcpyDeviceToHost));
  for (auto& r : rp) r &= ~3;
  if ((int64_t)occ_off[n_s1] != n_lidx2) return set_err(GSPX_ERR_INVALID, "pair tiles: occ_off does not end at n_lidx2");
  std::vector<int> hdr((size_t)nb * 8), desc((size_t)n_s1 * 2), src((size_t)n_lidx2);
  std::vector<uint16_t> ownpos((size_t)N, 0), l1((size_t)g->nnz_int), l2((size_t)n_lidx2);
  std::vector<int> owndesc((size_t)N * 2, 0);
  int n1max = 0, n2max = 0, e1max = 0, e2max = 0;
  for (int b = 0; b < nb; ++b) {
    const int lo = s1ptr[b], hi = s1ptr[b + 1];
    const int r0 = b * block_rows, r1 = (int)std::min<int64_t>((int64_t)r0 + block_rows, N);
    int found = 0;
    for (int o = lo; o < hi; ++o) {
      const int r = s1rows[o];
      if (r < 0 || r >= N) return set_err(GSPX_ERR_INVALID, "pair tiles: bad S1 row");
      const int len = rp[r + 1] - rp[r];
      const int64_t rel = (int64_t)occ_off[o] - (int64_t)occ_off[lo];
      if ((int64_t)occ_off[o + 1] - (int64_t)occ_off[o] != len) return set_err(GSPX_ERR_INVALID, "pair tiles: occ_off does not match the rows");
      if (len >= 256 || rel >= (1 << 24) || hi - lo >= 65535)
        return set_err(GSPX_ERR_INVALID, "pair tiles: a row of %d entries / a tile of %d rows does not fit the descriptors", len, hi - lo);
      desc[(size_t)o * 2] = r;
      desc[(size_t)o * 2 + 1] = (int)rel | (len << 24);
      for (int j = 0; j < len; ++j) src[(size_t)occ_off[o] + j] = rp[r] + j;
      if (r >= r0 && r < r1) {
        ownpos[r] = (uint16_t)(o - lo);
        owndesc[(size_t)r * 2] = (int)rel | (len << 24);
        owndesc[(size_t)r * 2 + 1] = rp[r] - rp[r0];
        ++found;
      }
    }
    if (found != r1 - r0) return set_err(GSPX_ERR_INVALID, "pair tiles: S1 must contain the block's own rows");
    int* h = &hdr[(size_t)b * 8];
    h[0] = lo;
    h[1] = hi - lo;
    h[2] = s2ptr[b];
    h[3] = s2ptr[b + 1] - s2ptr[b];
    h[4] = (int)occ_off[lo];
    h[5] = (int)(occ_off[hi] - occ_off[lo]);
    h[6] = rp[r0];
    h[7] = rp[r1] - rp[r0];
    if (h[3] >= 65535) return set_err(GSPX_ERR_INVALID, "pair tiles: S2 too large for 16-bit positions");
    n1max = std::max(n1max, h[1]);
    n2max = std::max(n2max, h[3]);
    e1max = std::max(e1max, h[5]);
    e2max = std::max(e2max, h[7]);
  }
  for (int64_t i = 0; i < g->nnz_int; ++i) l1[(size_t)i] = lidx1[i] == 0xFFFF ? 0 : lidx1[i];  // pads carry the value 0
  for (int64_t i = 0; i < n_lidx2; ++i) l2[(size_t)i] = lidx2[i] == 0xFFFF ? 0 : lidx2[i];
  const size_t esz = elt_size(g->dtype);
  CHK(cp.hdr.alloc(hdr.size() * 4 + 64));
  CHK(cp.desc.alloc(desc.size() * 4 + 64));
  CHK(cp.s2rows.alloc((size_t)n_s2 * 4 + 64));
  CHK(cp.lidx1.alloc((size_t)g->nnz_int * 2 + 64));
  CHK(cp.lidx2.alloc((size_t)n_lidx2 * 2 + 64));
  CHK(cp.src.alloc((size_t)n_lidx2 * 4 + 64));
  CHK(cp.val2.alloc((size_t)n_lidx2 * esz + 64));
  CHK(cp.ownpos.alloc((size_t)N * 2 + 64));
  CHK(cp.owndesc.alloc((size_t)N * 8 + 64));
  HIPCHK(hipMemcpy(cp.hdr.p, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.desc.p, desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.s2rows.p, s2rows, (size_t)n_s2 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.lidx1.p, l1.data(), (size_t)g->nnz_int * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.lidx2.p, l2.data(), (size_t)n_lidx2 * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.src.p, src.data(), (size_t)n_lidx2 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.ownpos.p, ownpos.data(), (size_t)N * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.owndesc.p, owndesc.data(), (size_t)N * 8, hipMemcpyHostToDevice));
  cp.n_s1 = n_s1;
  cp.n_s2 = n_s2;
  cp.rows = block_rows;
  cp.nb = nb;
  cp.n1max = n1max;
  cp.n2max = n2max;
  cp.e1max = e1max;
  cp.e2max = e2max;
  cp.total2 = n_lidx2;
  cp.val_lmax = -1.0;
  if (stats) {
    stats[0] = nb; stats[1] = n1max; stats[2] = n2max; stats[3] = e1max; stats[4] = e2max; stats[5] = n_lidx2;
  }
  return GSPX_OK;
}

template <typename T>
static int cheb_pair_dev_t(gspx_graph* g, double lmax, int M, const double* c, int64_t Nsig, const T* x, T* y,
                           int chunk_lanes) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  ChebPairTiles& cp = g->cp;
  const int N = (int)g->N, K = M - 1;
  constexpr unsigned TVEC = 16 / (unsigned)sizeof(T);
  const unsigned ld = (unsigned)Nsig;
  if ((ld % TVEC) != 0 || (((uintptr_t)x / sizeof(T)) % TVEC) != 0 || (((uintptr_t)y / sizeof(T)) % TVEC) != 0)
    return set_err(GSPX_ERR_INVALID, "pair filter: panels must be made of aligned 16-byte pieces");
  const size_t U = (size_t)N * ld;
  if (U * sizeof(T) >= ((size_t)1 << 31)) return set_err(GSPX_ERR_INVALID, "pair filter: panel of 2 GiB or more");
  CHK(ensure_factor<T>(g, lmax));
  if (cp.val_lmax != lmax) {
    hipLaunchKernelGGL((gspx::k_gather_vals<T>), dim3((unsigned)((cp.total2 + 255) / 256)), dim3(256), 0, st, g->fval.as<T>(),
                       cp.src.as<int>(), (size_t)cp.total2, cp.val2.as<T>());
    cp.val_lmax = lmax;
  }
  CHK(ctx->ws_t.ensure(4 * U * sizeof(T) + 256));
  CHK(ctx->ws_r.ensure(U * sizeof(T) + 256));
  T* X[4] = {ctx->ws_t.as<T>(), ctx->ws_t.as<T>() + U, ctx->ws_t.as<T>() + 2 * U, ctx->ws_t.as<T>() + 3 * U};
  T* R = ctx->ws_r.as<T>();
  const int* perm = g->has_perm ? g->perm.as<int>() : nullptr;
  size_t ev_idx = 0;
  hipEvent_t e0 = pool_event(ctx, ++ev_idx), e1 = pool_event(ctx, ++ev_idx), e2 = pool_event(ctx, ++ev_idx);
  if (!e0 || !e1 || !e2) return set_err(GSPX_ERR_HIP, "hipEventCreate failed");
  HIPCHK(hipEventRecord(e0, st));
  launch_permute_in<T>(x, ld, X[0], ld, N, perm, (int)TVEC, st);
  HIPCHK(hipEventRecord(e1, st));

  const int CW = chunk_lanes;
  gspx::ChebPairArgs<T> a{};
  a.rowptr = g->rptr.as<int>();
  a.hdr = cp.hdr.as<int>();
  a.desc = (const int2*)cp.desc.p;
  a.s2rows = cp.s2rows.as<int>();
  a.val2 = cp.val2.as<T>();
  a.lidx2 = cp.lidx2.as<gspx::u16>();
  a.lidx1 = cp.lidx1.as<gspx::u16>();
  a.ownpos = cp.ownpos.as<gspx::u16>();
  a.owndesc = (const int2*)cp.owndesc.p;
  a.s2rows_bytes = (unsigned)((size_t)cp.n_s2 * 4);
  a.desc_bytes = (unsigned)((size_t)cp.n_s1 * 8);
  a.owndesc_bytes = (unsigned)((size_t)N * 8);
  a.perm = perm;
  a.N = N;
  a.BR = cp.rows;
  a.nb = cp.nb;
  a.per_xcd = (cp.nb + 7) / 8;
  a.ld = ld;
  a.ldy = ld;
  a.y = y;
  a.panel_bytes = (unsigned)(U * sizeof(T));
  a.ncol = (int)(((size_t)ld * sizeof(T) + (size_t)CW * 16 - 1) / ((size_t)CW * 16));
  auto r16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t sz_h = (size_t)cp.n2max * CW * 16, sz_g = (size_t)cp.n1max * CW * 16;
  a.off_g = (int)sz_h;
  a.off_val = (int)(sz_h + sz_g);
  a.off_idx1 = (int)(a.off_val + r16((size_t)cp.e1max * sizeof(T)));
  a.off_idx2 = (int)(a.off_idx1 + r16((size_t)cp.e1max * 2));
  size_t lds = (size_t)a.off_idx2 + r16((size_t)cp.e2max * 2) + 64;
  typedef void (*kern_t)(const gspx::ChebPairArgs<T>);
  kern_t kern = CW == 2 ? (kern_t)gspx::k_cheb_pair<T, 2> : CW == 4 ? (kern_t)gspx::k_cheb_pair<T, 4> : CW == 8 ? (kern_t)gspx::k_cheb_pair<T, 8>
                                                                                                        : (kern_t)gspx::k_cheb_pair<T, 16>;
  // option "pair_kernel": 2 (default) the pipelined build (k_cheb_pair2) when one exists for this shape, 1 the first one
  if (ctx->opt.pair_kernel >= 2) {
    const int NG = 512 / CW;
    const int it1 = (cp.n1max + NG - 1) / NG, it2 = (cp.rows + NG - 1) / NG;
    kern_t k2 = nullptr;
#define GSPX_PAIR2(cw, i1, i2) \
    if (CW == cw && it1 <= i1 && it2 <= i2 && !k2) k2 = (kern_t)gspx::k_cheb_pair2<T, cw, i1, i2>;
    GSPX_PAIR2(2, 1, 1) GSPX_PAIR2(2, 2, 1)
    GSPX_PAIR2(4, 1, 1) GSPX_PAIR2(4, 2, 1) GSPX_PAIR2(4, 3, 2)
    GSPX_PAIR2(8, 2, 1) GSPX_PAIR2(8, 3, 1) GSPX_PAIR2(8, 4, 2)
#undef GSPX_PAIR2
    static const bool dbg_on = getenv("GSPX_PAIR_DEBUG") != nullptr;
    if (k2 && dbg_on && std::is_same<T, double>::value) {
      if (k2 == (kern_t)gspx::k_cheb_pair2<T, 8, 3, 1>) k2 = (kern_t)gspx::k_cheb_pair2<T, 8, 3, 1, true>;
      if (k2 == (kern_t)gspx::k_cheb_pair2<T, 4, 2, 1>) k2 = (kern_t)gspx::k_cheb_pair2<T, 4, 2, 1, true>;
    }
    if (k2) {
      kern = k2;
      a.off_ms2 = (int)r16(lds);
      a.off_ms1 = (int)(a.off_ms2 + r16((size_t)cp.n2max * 4));
      a.off_mown = (int)(a.off_ms1 + r16((size_t)cp.n1max * 8));
      a.meta_bytes = (int)(a.off_mown + r16((size_t)cp.rows * 8) - a.off_ms2);
      lds = (size_t)a.off_ms2 + 2 * (size_t)a.meta_bytes + 64;
      a.val2_bytes = (unsigned)std::min<size_t>((size_t)cp.total2 * sizeof(T) + 64, ((size_t)1 << 31) - 1);
      a.lidx2_bytes = (unsigned)std::min<size_t>((size_t)cp.total2 * 2 + 64, ((size_t)1 << 31) - 1);
      a.lidx1_bytes = (unsigned)std::min<size_t>((size_t)g->nnz_int * 2 + 64, ((size_t)1 << 31) - 1);
      if ((size_t)cp.total2 * sizeof(T) >= ((size_t)1 << 31))
        return set_err(GSPX_ERR_INVALID, "pair filter: the gathered matrix values exceed 2 GiB");
    }
  }
  if (lds > (size_t)160 * 1024) return set_err(GSPX_ERR_INVALID, "pair filter: tiles need %zu bytes of LDS", lds);
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 1;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, lds));
  per_cu = std::max(1, std::min(per_cu, 4));
  if (ctx->opt.pair_workgroups_per_cu > 0) per_cu = (int)std::min<int64_t>(per_cu, ctx->opt.pair_workgroups_per_cu);
  unsigned nwg = (unsigned)std::max<int64_t>(8, ((int64_t)per_cu * ctx->cu_count) / 8 * 8);
  // option "pair_workgroups" (absolute; round 5): a small grid makes every workgroup walk many blocks - the shape of
  // the full-size runs - on a graph of a few thousand vertices (tools/pair_ladder.py)
  if (ctx->opt.pair_workgroups > 0)
    nwg = (unsigned)std::max<int64_t>(8, std::min<int64_t>(ctx->opt.pair_workgroups, (int64_t)nwg) / 8 * 8);
  nwg = std::min(nwg, 8u * (unsigned)std::max(a.per_xcd, 1));
  ctx->timing[7] = (double)nwg;
  ctx->timing[5] = (double)lds;
  ctx->timing[6] = (double)per_cu;
  a.dbg = nullptr;
  if (getenv("GSPX_PAIR_DEBUG")) {
    CHK(ctx->ws_w.ensure((size_t)nwg * 64 + 64));
    HIPCHK(hipMemsetAsync(ctx->ws_w.p, 0, (size_t)nwg * 64, st));
    a.dbg = (unsigned long long*)ctx->ws_w.p;
  }

  // pair j = steps 2j - 1 and 2j.  Panels: P = T_{2j-2}, Q = T_{2j-3}; A = T_{2j-1}, B = T_{2j} go to the two free ones
  int iP = 0, iQ = -1, iA = 1, iB = 2, free2 = 3;
  const int npairs = K / 2;
  int last_flushed = 0;  // highest order folded into R so far (pair 1 folds T_0 too)
  int launches = 0;
  for (int j = 1; j <= npairs; ++j) {
    a.P = X[iP];
    a.Q = iQ >= 0 ? X[iQ] : nullptr;
    a.A = X[iA];
    a.B = X[iB];
    a.R = R;
    if (j == 1) {  // T_1 = (L - a2) x / a1 = F x / 2;  T_2 = F T_1 - T_0
      a.sA = T(0.5); a.gA = T(0); a.sB = T(1); a.gB = T(-1);
    } else {
      a.sA = T(1); a.gA = T(-1); a.sB = T(1); a.gB = T(-1);
    }
    const bool final = j == npairs;
    const bool flush = j == 1 || final || ((j & 1) == 1);
    a.flush = 0;
    a.final = 0;
    a.wB = a.wA = a.wP = a.wQ = T(0);
    if (flush) {
      a.flush = j == 1 ? 1 : 2;
      a.final = final ? 1 : 0;
      a.wB = (T)c[2 * j];
      a.wA = (T)c[2 * j - 1];
      if (j == 1) a.wP = (T)(0.5 * c[0]);
      else if (last_flushed < 2 * j - 2) {  // the previous pair was not flushed: its two terms too
        a.wP = (T)c[2 * j - 2];
        a.wQ = (T)c[2 * j - 3];
      }
      last_flushed = 2 * j;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, a);
    ++launches;
    // rotate: the new P is B, the new Q is A; the old P and Q panels are free
    const int oP = iP, oQ = iQ >= 0 ? iQ : free2;
    iP = iB;
    iQ = iA;
    iA = oQ;
    iB = oP;
  }
  HIPCHK(hipEventRecord(e2, st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  if (a.dbg) {  // cycles of wave 0 of every workgroup in the LAST launch, averaged per pass
    std::vector<unsigned long long> h((size_t)nwg * 8);
    HIPCHK(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned w = 0; w < nwg; ++w)
      for (int i = 0; i < 8; ++i) sum[i] += (double)h[(size_t)w * 8 + i];
    const double np = std::max(sum[6], 1.0);
    fprintf(stderr, "pair debug (cycles per pass, wave 0, %u workgroups, %.0f passes): tile wait %.0f | barrier A %.0f | phase 1 %.0f | "
                    "barrier B %.0f | phase 2 %.0f | issue %.0f | block change (per pass) %.0f\n", nwg, np, sum[0] / np, sum[1] / np,
            sum[2] / np, sum[3] / np, sum[4] / np, sum[5] / np, sum[7] / np);
  }
  float f01 = 0, f12 = 0;
  HIPCHK(hipEventElapsedTime(&f01, e0, e1));
  HIPCHK(hipEventElapsedTime(&f12, e1, e2));
  ctx->timing[0] = f01 + f12;
  ctx->timing[1] = f12;
  ctx->timing[2] = launches;
  ctx->timing[3] = f01;
  ctx->timing[4] = ctx->timing[7];  // (experimental: the grid size, where the default path reports combine_ms)
  return GSPX_OK;
}

extern "C" int gspx_cheby_pair_filter_dev(gspx_graph* g, double lmax, int M, const double* coeffs, int64_t Nsig,
                                          const void* x_dev, void* y_dev, int chunk_lanes, double* kernel_ms) {
  if (g) replay_reset(g->ctx);
  CHK(check_filter_args(g, lmax, 1, M, coeffs, Nsig, x_dev, y_dev, GSPX_ANALYSIS));
  if (!g->cp.rows) return set_err(GSPX_ERR_INVALID, "pair filter: the graph carries no pair tiles (gspx_graph_set_cheb_pair_tiles)");
  if (((M - 1) & 1) != 0) return set_err(GSPX_ERR_INVALID, "pair filter: the order must be even (two orders per launch)");
  if (chunk_lanes != 2 && chunk_lanes != 4 && chunk_lanes != 8 && chunk_lanes != 16)
    return set_err(GSPX_ERR_INVALID, "pair filter: chunk_lanes must be 2, 4, 8 or 16");
  if (kernel_ms) *kernel_ms = 0;
  if (Nsig == 0 || g->N == 0) return GSPX_OK;
  HIPCHK(hipSetDevice(g->ctx->device));
  int rc = g->dtype == GSPX_F32 ? cheb_pair_dev_t<float>(g, lmax, M, coeffs, Nsig, (const float*)x_dev, (float*)y_dev, chunk_lanes)
                                : cheb_pair_dev_t<double>(g, lmax, M, coeffs, Nsig, (const double*)x_dev, (double*)y_dev, chunk_lanes);
  if (rc == GSPX_OK && kernel_ms) *kernel_ms = g->ctx->timing[0];
  return rc;
}
