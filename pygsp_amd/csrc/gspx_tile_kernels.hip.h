// gspx_tile_kernels.hip.h - the recurrence step with the gathered panel staged in LDS ("x tile").
//
//   T_k = scale * (F T_{k-1}) + gamma * T_{k-2}   (+ the fused flush of one filter)
//   approximations.py:107-109, same arithmetic as k_step_panel / k_step_lds (gspx_kernels.hip.h)
//
// The internal vertex order is cut into 64-row blocks.  The host (pygsp_amd/tiling.py) lists, per
// block, the distinct rows its entries touch (S1: 123 rows on average on the headline graph) and,
// for every stored entry, the 8-bit position of its column inside that list (a staged tile has at most
// 160 rows).  A workgroup
//   1. loads the S1 rows of T_{k-1} straight into LDS (buffer_load ... lds: 1 KiB per wave
//      instruction, no VGPRs), one 256-byte column chunk at a time;
//   2. copies the block's contiguous slice of matrix entries (value + 8-bit position) into LDS,
//      coalesced, once for all column chunks of the block;
//   3. computes its rows from LDS (ds_read_b128 gathers, broadcast reads of the entries).
// Against the plain gather kernels this turns 11 gathers per row (a third of which miss the L1 and
// re-fetch from L2) into 1.9 coalesced row fetches per row, and shrinks the streamed matrix from
// (elt + 4) to (elt + 1) bytes per entry.  Workgroups are persistent; XCD x walks a contiguous
// eighth of the blocks, its workgroups interleaved, so the halo rows of concurrently staged blocks
// meet in that XCD's L2.  Block headers are prefetched two blocks ahead, row lists one block ahead.
// Blocks whose tile does not fit (n1 > 160 rows or rows longer than the LDS slice) take a plain
// global-gather path inside the same kernel.
#pragma once

#include "gspx_kernels.hip.h"

namespace gspx {

template <typename T> struct TileArgs {
  const int* rowptr;   // internal padded CSR
  const int* col;      // global columns (slow path only)
  const T* val;        // factor values F
  const int* hdr;      // [nb][4]: s1lo, n1 (-1: slow path), rp0 (first entry of the block), ent (entries)
  const int* s1rows;   // concatenated S1 lists
  const u8* lidx;      // [nnz_int] position of each entry's column in its block's S1 (< 256; pads: 0)
  const T* cur;
  const T* old;
  const int* old_rows;  // OLDNAT builds: row of `old` holding T_{k-2} of internal row r (the caller's x, unpermuted)
  T* out;
  T* racc;
  T* y;
  const int* perm;
  int N;
  u32 ld, ldy;
  u32 panel_bytes, val_bytes, lidx_bytes;
  int nb, ncol, per_xcd;
  int lds_bytes;
  T scale, gamma;
  T beta;        // coefficient of T_{k-1}'s own row (Newton-form steps; 0 for the recurrence)
  T wn, wc, wo;  // flush weights of the single filter
  // extra input panels summed into the row: out += sum_f wts[f] * inp[f][row] (synthesis by Clenshaw)
  const T* inp;  // [nin][N][ld]
  const T* wts;  // device, [nin]
  int nin;
  int flush;     // 0 none, 1 write, 2 accumulate
  int final;     // 1: the flush result goes to y (caller's order)
  int reverse;    // 1: every XCD walks its block range from the end (odd steps: the tail of the previous
                  // step's panels is still in the Infinity Cache)
  int nt;         // non-temporal accesses: bit 0 matrix entries, bit 1 accumulator, bit 2 T_{k-2} rows,
                  // bit 3 T_k stores (keeps streamed data out of the Infinity Cache)
};

constexpr int GSPX_TILE_BR = 64;      // rows per block
constexpr int GSPX_TILE_MAXN1 = 160;  // S1 rows a workgroup stages (5 per group)

// LG = lanes per row group: 16 (256-byte column chunks; 32 groups x 2 rows) or, for narrow panels, 8 / 4 / 2 / 1
// (128- / 64- / 32- / 16-byte rows; 64 groups x 1 row, NT = 64 LG threads per workgroup: every lane has a piece of a
// row, and the smaller workgroups keep more blocks in flight per CU - a narrow pass moves little data per
// memory round trip).  NCOL = 1, 2: that many column chunks per row; 0: a.ncol chunks
// OLDNAT: T_{k-2} rows are read through a.old_rows (step 2 of a filter whose input panel was not
// copied into the internal order first)
// Software pipelining of a pass (round 2; measured per panel shape on one box against the previous build,
// profiles/r02_tile_variants*.log - the kernel sits at the 128-VGPR budget of two workgroups per CU, so what
// wins is also a question of which order does not spill):
//   PF          the pass's T_{k-2} / accumulator rows and a new block's matrix entries are loaded one pass
//               ahead, next to its tile; the previous pass's stores are issued before them and the tile's
//               LDS-DMA loads LAST, so at the first barrier of a pass the youngest thing a wave can be waiting
//               for is its own tile;
//   META_AFTER  the next block's row lists and header are loaded after the first barrier instead of before
//               it (the 4-chunk build only: before the barrier it spills);
//   otherwise they are loaded before the barrier and the wait in front of it skips exactly them
//               (s_waitcnt vmcnt(NMETA): VMEM retires in order, they are the NMETA youngest operations).
// fp64 x 64 signals +1.6 ... +2.8 % against the previous build, fp64 x 128 +9 %, narrow / single-chunk panels
// -0.8 ... -1.4 %: the previous build did not wait for a wave's own tile at all (see the first barrier below).
// The OLDNAT builds (one launch per call) keep the plain order and a full wait.
// Narrow panels (8-lane groups) issue the tile first and wait for everything: measured better there.
template <typename T, int NCOL, int LG, bool OLDNAT> struct TileSchedule {
  static constexpr bool PF = !OLDNAT;
  static constexpr bool TILE_LAST = PF && LG == 16;
  static constexpr bool PF_ENTRIES = PF;  // (leaving the entries with their own pass measured 2-3 % slower)
  static constexpr bool META_AFTER = !OLDNAT && LG == 16 && NCOL == 0;
};

// INS: the step adds extra input panels to the row (synthesis by Clenshaw, a.nin > 0) - its own build, so
// that the analysis path does not carry their registers across the row products
// (A bound of five waves per SIMD for the 4-lane build - five resident workgroups instead of four - costs 20-50
// bytes of scratch and 20 % of the fp64 build's speed: measured, dropped.)
// CL: lanes of a row group in the COMPUTE phases (row products, T_{k-2} / accumulator loads, stores), when a row
// has fewer 16-byte pieces than the LG lanes its tile rows are staged with (rows of 80 - 240 bytes that are not a
// power of two: 10 / 12 / 24 fp64 signals ...).  The LDS-DMA keeps its LG-lane layout (a wave instruction writes
// 64 / LG tile rows of 16 LG bytes, the lanes beyond the row switched off by the bounds check); the compute phases
// regroup the workgroup's NT = CL x (groups) threads into CL-lane groups, so every lane holds a piece of a row -
// with LG-lane groups a quarter to three eighths of the lanes idled through every row product.  CL == LG: as before.
// (two resident workgroups of 320 / 384 threads are 10 / 12 waves per CU, three per SIMD: those builds may use 168
// registers; their longer row lists per staging group - 8 / 7 instead of 5 - would spill at 128)
// (128-row blocks for the 8-lane build - twice the bytes per pass against the same two barriers - measured
// -3.7 ... +0.8 % in round 5, profiles/r05_narrow_rows.md; retired)
// MIX (calibration builds, gspx_bench_step_mix - never part of a filter call): the same pass with the ARITHMETIC REMOVED -
// same grid and walk, same LDS-DMA tile loads of the same S1 lists, same T_{k-2} / accumulator loads, same entry
// stream into LDS, same stores with the same cache bits; the row products (the per-entry LDS gathers and FMAs) are
// replaced by one tile read per row.  1: both barriers of a pass kept (what the memory system delivers to this
// access mix behind this synchronisation); 2: the barriers dropped too (the access mix alone; tile rows may be read
// while still arriving - the values mean nothing in either build).
template <typename T, int NCOL, int LG = 16, bool OLDNAT = false, bool INS = false, int NT = 512, int CL = LG,
          int MIX = 0>
__global__ __launch_bounds__(NT, 4) void k_step_tile(const TileArgs<T> a) {
  static_assert(NT == 512 || (NT == 64 * LG && NCOL == 1) || (CL < LG && NCOL == 1 && NT % 64 == 0 && NT % CL == 0),
                "narrow builds: one row per group, one chunk per row");
  constexpr int BR = GSPX_TILE_BR;
  static_assert(CL <= LG && (BR % (NT / CL)) == 0, "compute groups must tile the block's rows");
  constexpr int MAXN1 = GSPX_TILE_MAXN1;
  constexpr bool PF = TileSchedule<T, NCOL, LG, OLDNAT>::PF;
  constexpr bool META_AFTER = TileSchedule<T, NCOL, LG, OLDNAT>::META_AFTER;
  constexpr bool TILE_LAST = TileSchedule<T, NCOL, LG, OLDNAT>::TILE_LAST;
  constexpr bool PF_ENTRIES = TileSchedule<T, NCOL, LG, OLDNAT>::PF_ENTRIES;
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef typename VT<T, VEC>::t V;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) unsigned char gspx_smem[];

  const int tid = threadIdx.x;
  constexpr int NGD = NT / LG;             // staging (LDS-DMA) groups per workgroup: LG lanes each
  constexpr int NG = NT / CL;              // compute row groups per workgroup: CL lanes each
  constexpr int NE = (512 + NT - 1) / NT;  // 16-byte pieces of matrix values a thread prefetches
  constexpr int RPG = BR / NG;   // rows of the block per compute group
  constexpr int ST = (MAXN1 + NGD - 1) / NGD;  // tile rows a staging group loads
  constexpr int RB = LG * 16;              // bytes of a tile row
  const int lane_d = tid & (LG - 1);       // staging lane / group
  const int grp_d = tid / LG;
  const int lane16 = CL == LG ? lane_d : tid % CL;  // compute lane / group
  const int grp = CL == LG ? grp_d : tid / CL;
  const int wave = tid >> 6;
  const int nwx = (int)(gridDim.x >> 3);
  const int xlo = (int)(blockIdx.x & 7) * a.per_xcd;
  int k1 = xlo + a.per_xcd;
  if (k1 > a.nb) k1 = a.nb;
  const int k0 = xlo + (int)(blockIdx.x >> 3);
  if (k0 >= k1) return;

  constexpr u32 POISON = 0x80000000u;
  const rsrc_t rcur = __builtin_amdgcn_make_buffer_rsrc((void*)a.cur, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rold = __builtin_amdgcn_make_buffer_rsrc((void*)a.old, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rra = __builtin_amdgcn_make_buffer_rsrc((void*)a.racc, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)a.val, 0, a.val_bytes, 0x00020000);
  const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)a.lidx, 0, a.lidx_bytes, 0x00020000);
  const u32 ldb = a.ld * (u32)sizeof(T);

  struct Meta { int rows[ST]; int rp[RPG + 1]; int orow[OLDNAT ? RPG : 1]; };
  // walk position -> block: the XCD's range front to back, or back to front
  const int xflip = xlo + k1 - 1;
  const bool rev = a.reverse != 0;
  auto phys = [&](int p) { return rev ? xflip - p : p; };
  auto load_hdr = [&](int p) { return *(const int4*)(a.hdr + (size_t)phys(p) * 4); };
  auto uniform = [](int4 h) {
    int4 u;
    u.x = __builtin_amdgcn_readfirstlane(h.x); u.y = __builtin_amdgcn_readfirstlane(h.y);
    u.z = __builtin_amdgcn_readfirstlane(h.z); u.w = __builtin_amdgcn_readfirstlane(h.w);
    return u;
  };
  auto load_meta = [&](int k, const int4& h) {
    Meta m;
    const int n1 = h.y;
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      const int u = grp_d + NGD * t;
      m.rows[t] = a.s1rows[h.x + (u < n1 ? u : 0)];
    }
    int r = phys(k) * BR + grp * RPG;
#pragma unroll
    for (int t = 0; t < RPG + 1; ++t) m.rp[t] = a.rowptr[(r + t) <= a.N ? (r + t) : a.N];
    if constexpr (OLDNAT) {
#pragma unroll
      for (int t = 0; t < RPG; ++t) m.orow[t] = a.old_rows[(r + t) < a.N ? (r + t) : a.N - 1];
    }
    return m;
  };
  V* const tile = (V*)gspx_smem;
  auto stage = [&](const Meta& m, int n1, u32 cb) {
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      if (grp_d + NGD * t < n1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rcur, (lds_ptr)(gspx_smem + (wave * (64 / LG) + NGD * t) * RB), 16,
                                                 (u32)m.rows[t] * ldb + cb, 0, 0, 0);
    }
  };
  auto chunk_off_d = [&](int c) {  // the staging lane's piece of column chunk c
    const u32 col0 = (c * LG + lane_d) * VEC;
    return col0 < a.ld ? col0 * (u32)sizeof(T) : POISON;
  };
  auto chunk_off = [&](int c) {    // the compute lane's
    const u32 col0 = (c * LG + lane16) * VEC;
    return col0 < a.ld ? col0 * (u32)sizeof(T) : POISON;
  };

  // Loads that a pass consumes are issued one pass ahead, together with its tile: T_{k-2} (and the
  // accumulator) of the group's rows, and - for the first pass of a block - the block's slice of matrix
  // entries.  By the time the pass reaches its first barrier they are as old as the tile loads, so the
  // barrier does not expose a fresh memory round trip every pass.
  V ov[RPG], ra[RPG];
  u32x4 ev[NE], ei = 0;
#pragma unroll
  for (int q = 0; q < NE; ++q) ev[q] = 0;
  auto prefetch_rows = [&](const Meta& m, int blk, int c) {
    const int r0 = phys(blk) * BR + grp * RPG;
    const u32 cb = chunk_off(c);
#pragma unroll
    for (int t = 0; t < RPG; ++t) {
      const u32 off = (r0 + t < a.N && cb != POISON) ? (u32)(r0 + t) * ldb + cb : POISON;
      u32 oo = a.gamma != T(0) ? off : POISON;
      if constexpr (OLDNAT) oo = (off != POISON && a.gamma != T(0)) ? (u32)m.orow[t] * ldb + cb : POISON;
      if (a.nt & 4) ov[t] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rold, oo, 0, 2));
      else ov[t] = VT<T, VEC>::bload(rold, oo);
      const u32 ro = a.flush == 2 ? off : POISON;
      if (a.nt & 2) ra[t] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rra, ro, 0, 2));
      else ra[t] = VT<T, VEC>::bload(rra, ro);
    }
  };
  auto prefetch_entries = [&](const int4& h) {  // the block's slice of matrix entries, coalesced 16-byte pieces
    const int nv16 = (h.w * (int)sizeof(T) + 15) >> 4, ni16 = (h.w + 15) >> 4;
    const u32 io = tid < ni16 ? (u32)h.z + tid * 16u : POISON;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int i = tid + q * NT;
      const u32 vo = i < nv16 ? (u32)h.z * (u32)sizeof(T) + i * 16u : POISON;
      if (a.nt & 1) ev[q] = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 2);
      else ev[q] = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0);
    }
    if (a.nt & 1) ei = __builtin_amdgcn_raw_buffer_load_b128(ri, io, 0, 2);
    else ei = __builtin_amdgcn_raw_buffer_load_b128(ri, io, 0, 0);
  };

  int4 H = uniform(load_hdr(k0));
  Meta M = load_meta(k0, H);
  // the workgroup's blocks (a static, strided walk of its XCD's range): k now, kn next (row lists
  // prefetched), knn after that (header prefetched)
  int k = k0, kn = k0 + nwx, knn = k0 + 2 * nwx;
  int4 Hn = uniform(load_hdr(kn < k1 ? kn : k0));
  if constexpr (!TILE_LAST) {
    if (H.y >= 0) stage(M, H.y, chunk_off_d(0));
  }
  if constexpr (PF) prefetch_rows(M, k0, 0);
  if constexpr (PF_ENTRIES) {
    if (H.y >= 0) prefetch_entries(H);
  }
  if constexpr (TILE_LAST) {  // the tile last, as inside the loop
    __builtin_amdgcn_sched_barrier(0);
    if (H.y >= 0) stage(M, H.y, chunk_off_d(0));
    __builtin_amdgcn_sched_barrier(0);
  }

  // one pass = column chunk c of block k; returns false after the workgroup's last pass
  auto pass = [&](const int c, const bool first, const bool last) __attribute__((always_inline)) {
    const int n1 = H.y, rp0 = H.z, ent = H.w;
    const bool fast = n1 >= 0;
    T* const mval = (T*)(tile + (fast ? n1 : 0) * LG);
    u8* const midx = (u8*)(mval + ent);
    const int row0 = phys(k) * BR + grp * RPG;
    int rs[RPG + 1];
#pragma unroll
    for (int t = 0; t < RPG + 1; ++t) rs[t] = M.rp[t] & ~3;
    const u32 col0 = (c * LG + lane16) * VEC;
    const bool on = col0 < a.ld;
    const u32 cb = on ? col0 * (u32)sizeof(T) : POISON;
    if constexpr (!PF) prefetch_rows(M, k, c);
    V ins[INS ? RPG : 1];
#pragma unroll
    for (int t = 0; t < (INS ? RPG : 1); ++t) ins[t] = 0;
    if constexpr (INS) {
      const rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.inp, 0, a.panel_bytes * (u32)a.nin, 0x00020000);
      for (int f = 0; f < a.nin; ++f) {
        const T w = a.wts[f];
#pragma unroll
        for (int t = 0; t < RPG; ++t) {
          const u32 off = (row0 + t < a.N) ? (u32)(row0 + t) * ldb + cb : POISON;
          ins[t] += w * VT<T, VEC>::bload(rin, off == POISON ? POISON : off + (u32)f * a.panel_bytes);
        }
      }
    }
    if constexpr (!PF_ENTRIES) {
      if (first && fast) prefetch_entries(H);
    }
    int4 Hv = Hn;
    if constexpr (!META_AFTER) {
      if (last) {  // the next block's row lists, the header after that
        M = load_meta(kn < k1 ? kn : k, Hn);
        Hv = load_hdr(knn < k1 ? knn : k);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (first && fast) {  // the entries (fetched one pass ahead) go to LDS, once for all chunks of the block
      const int nv16 = (ent * (int)sizeof(T) + 15) >> 4, ni16 = (ent + 15) >> 4;
#pragma unroll
      for (int q = 0; q < NE; ++q)
        if (tid + q * NT < nv16) *(u32x4*)((unsigned char*)mval + (tid + q * NT) * 16) = ev[q];
      if (tid < ni16) *(u32x4*)((unsigned char*)midx + tid * 16) = ei;
      for (int i = tid + NE * NT; i < nv16; i += NT)  // slices longer than 8 KiB of values: rare
        *(u32x4*)((unsigned char*)mval + i * 16) =
            __builtin_amdgcn_raw_buffer_load_b128(rv, (u32)rp0 * (u32)sizeof(T) + i * 16u, 0, 0);
      for (int i = tid + NT; i < ni16; i += NT)
        *(u32x4*)((unsigned char*)midx + i * 16) =
            __builtin_amdgcn_raw_buffer_load_b128(ri, (u32)rp0 + i * 16u, 0, 0);
    }
    // Every wave waits for ITS OWN tile loads before the barrier: `buffer_load ... lds` is tracked by vmcnt only,
    // and __syncthreads() lowers to `s_waitcnt lgkmcnt(0); s_barrier` - it does not wait for them.  (Until
    // round 2 the only vmcnt wait before this barrier came from the entry writes above, which waves without
    // an entry to write skip: their tile rows could be read by the other waves while still in flight.)
    // Everything outstanding here was issued one pass ago and is needed now, so vmcnt(0) costs no extra wait.
    if constexpr (TILE_LAST && !META_AFTER) {
      // the tile DMA was the last thing issued one pass ago; the only VMEM operations younger than it are the
      // NMETA loads of the next block's row lists / header issued above in a block's last pass (and the extra
      // input panels of a synthesis step, which make the count data dependent: full wait then)
      constexpr int NMETA = ST + RPG + 1 + 1;
      static_assert(NMETA < 16, "vmcnt immediate");
      // (entries loaded in this pass were issued before the row lists: they are older, hence covered)
      if (last && !INS) __builtin_amdgcn_s_waitcnt(0x0070 | NMETA);  // vmcnt(NMETA) lgkmcnt(0)
      else __builtin_amdgcn_s_waitcnt(0x0070);
    } else {
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    }
    if constexpr (MIX != 2) __syncthreads();  // tile and entries in place, T_{k-2} / accumulator rows in registers
    if constexpr (META_AFTER) {
      if (last) {  // the next block's row lists and the header after that: in flight during the row products
        M = load_meta(kn < k1 ? kn : k, Hn);
        Hv = load_hdr(knn < k1 ? knn : k);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    V nv[RPG], res[RPG];
#pragma unroll
    for (int t = 0; t < RPG; ++t) {
      const int s = rs[t], e = rs[t + 1];
      V acc = 0, self = 0;
      if (MIX && fast) {  // calibration: one tile row and one entry per row, no products over the entries
        self = tile[(int)midx[s - rp0] * LG + lane16];
        acc = mval[s - rp0] * self;
      } else if (fast) {
        acc = lds_row_dot<T, V, LG, u8>(mval + (s - rp0), midx + (s - rp0), row0 + t < a.N ? e - s : 0, tile, lane16,
                                        self);
      } else if (row0 + t < a.N) {  // plain gathers from global memory (tile too large for LDS)
        for (int j = s; j < e; ++j) {
          const int cc = a.col[j];
          const V xv = VT<T, VEC>::bload(rcur, cc < a.N ? (u32)cc * ldb + cb : POISON);
          if (j == s) self = xv;  // entry 0 is the diagonal slot
          acc += a.val[j] * xv;
        }
      }
      nv[t] = a.scale * acc + a.gamma * ov[t] + a.beta * self;
      if constexpr (INS) nv[t] += ins[t];
      // the flush is formed here, while T_{k-1} (self) and T_{k-2} (ov) are at hand: after the barrier
      // their registers receive the next pass's rows
      res[t] = a.wn * nv[t] + a.wc * self + a.wo * ov[t];
      if (a.flush == 2) res[t] += ra[t];
    }
    // everybody is done with the tile (their LDS reads were consumed above).  A bare barrier: the
    // fence of __syncthreads() would also wait for the prefetches still in flight.
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
    if constexpr (MIX != 2) __builtin_amdgcn_s_barrier();
    bool more = true;
    if (last) {
      const int4 Hnn = uniform(Hv);  // youngest load so far: the row lists before it have landed too
      H = Hn;
      Hn = Hnn;
      more = kn < k1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!PF) {  // next pass's tile: the next chunk of this block, or chunk 0 of the next block
      if (more && H.y >= 0) stage(M, H.y, chunk_off_d(last ? 0 : c + 1));
    }
    // this pass's results (with PF: before the loads below, so that they are out of the way of the next
    // barrier's wait), ...
#pragma unroll
    for (int t = 0; t < RPG; ++t) {
      const int row = row0 + t;
      if (row < a.N && on) {
        if (a.final && !a.flush) {  // Newton form: the last step's result is the output
          const size_t orow = a.perm ? (size_t)a.perm[row] : (size_t)row;
          *(V*)(a.y + orow * a.ldy + col0) = nv[t];
          continue;
        }
        if (a.nt & 8) __builtin_nontemporal_store(nv[t], (V*)(a.out + (size_t)row * a.ld + col0));
        else *(V*)(a.out + (size_t)row * a.ld + col0) = nv[t];
        if (a.flush) {
          if (a.final) {
            const size_t orow = a.perm ? (size_t)a.perm[row] : (size_t)row;
            *(V*)(a.y + orow * a.ldy + col0) = res[t];
          } else {
            if (a.nt & 2) __builtin_nontemporal_store(res[t], (V*)(a.racc + (size_t)row * a.ld + col0));
            else *(V*)(a.racc + (size_t)row * a.ld + col0) = res[t];
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ... then everything the next pass consumes: its tile (the next chunk of this block, or chunk 0 of the
    // next block), its T_{k-2} / accumulator rows, and a new block's matrix entries
    if constexpr (PF) {
      if (more) {
        const int nc = last ? 0 : c + 1;
        if constexpr (!TILE_LAST) {
          if (H.y >= 0) stage(M, H.y, chunk_off_d(nc));
        }
        prefetch_rows(M, last ? kn : k, nc);
        if constexpr (PF_ENTRIES) {
          if (last && H.y >= 0) prefetch_entries(H);
        }
        if constexpr (TILE_LAST) {  // the tile last: see the first barrier
          __builtin_amdgcn_sched_barrier(0);
          if (H.y >= 0) stage(M, H.y, chunk_off_d(nc));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (last) {
      k = kn;
      kn = knn;
      knn += nwx;
    }
    return more;
  };
  for (;;) {
    if constexpr (NCOL == 1) {
      if (!pass(0, true, true)) break;
    } else if constexpr (NCOL == 2) {
      pass(0, true, false);
      if (!pass(1, false, true)) break;
    } else {
      bool more = true;
      for (int c = 0; c < a.ncol; ++c) more = pass(c, c == 0, c == a.ncol - 1);
      if (!more) break;
    }
  }
}

}  // namespace gspx

// ---------------------------------------------------------------------------------------------
// Gather tiles built on the device (what pygsp_amd/tiling.py computes with numpy):
// pass A: per 64-row block, the sorted distinct columns of its entries -> tmp[b][0..n1), n1[b]
// pass B: (after a scan of n1) s1rows, per-entry 16-bit positions, block headers
// ---------------------------------------------------------------------------------------------
namespace gspx {

constexpr int GSPX_TILE_SORTCAP = 4096;  // entries of a block the LDS sort holds
constexpr int GSPX_TILE_TMPCAP = 256;    // distinct rows kept per block (more: the block is "slow")

__global__ __launch_bounds__(256) void k_tiles_unique(const int* __restrict__ rowptr,
                                                      const int* __restrict__ col, int N, int nb,
                                                      int* __restrict__ tmp, int* __restrict__ n1) {
  constexpr int BR = GSPX_TILE_BR;
  __shared__ unsigned keys[GSPX_TILE_SORTCAP];
  __shared__ int wsum[4];
  const int b = blockIdx.x;
  const int r0 = b * BR, r1 = min(r0 + BR, N);
  const int e0 = rowptr[r0] & ~3, e1 = rowptr[r1] & ~3;
  const int ent = e1 - e0;
  if (ent > GSPX_TILE_SORTCAP) {  // a hub: no tile for this block
    if (threadIdx.x == 0) n1[b] = GSPX_TILE_TMPCAP + 1;
    return;
  }
  int n = 1;
  while (n < ent) n <<= 1;
  for (int i = threadIdx.x; i < n; i += 256) {
    unsigned v = 0xFFFFFFFFu;
    if (i < ent) {
      const int c = col[e0 + i];
      if (c < N) v = (unsigned)c;  // pads (col == N) sort to the end with the filler
    }
    keys[i] = v;
  }
  __syncthreads();
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned a = keys[i], c = keys[l];
          const bool up = (i & k) == 0;
          if ((a > c) == up) {
            keys[i] = c;
            keys[l] = a;
          }
        }
      }
      __syncthreads();
    }
  // distinct values, in order: position = number of "first occurrences" before it
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool first = i < n && keys[i] != 0xFFFFFFFFu && (i == 0 || keys[i] != keys[i - 1]);
    // block-wide exclusive scan of `first`
    const unsigned long long m = __ballot(first);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (first && off + before < GSPX_TILE_TMPCAP) tmp[(size_t)b * GSPX_TILE_TMPCAP + off + before] = (int)keys[i];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    base += total;
  }
  if (threadIdx.x == 0) n1[b] = base;
}

__global__ __launch_bounds__(256) void k_tiles_fill(const int* __restrict__ rowptr,
                                                    const int* __restrict__ col, int N, int nb,
                                                    const int* __restrict__ tmp, const int* __restrict__ n1,
                                                    const int* __restrict__ s1lo, int esz, int lds_bytes,
                                                    int* __restrict__ s1rows, u8* __restrict__ lidx,
                                                    int* __restrict__ hdr, int* __restrict__ nslow) {
  constexpr int BR = GSPX_TILE_BR, MAXN1 = GSPX_TILE_MAXN1, ROWB = 256;
  const int b = blockIdx.x;
  const int r0 = b * BR, r1 = min(r0 + BR, N);
  const int e0 = rowptr[r0] & ~3, e1 = rowptr[r1] & ~3;
  const int ent = e1 - e0;
  const int n = n1[b];
  const long need = (long)n * ROWB + (((long)ent * esz + 15) & ~15L) + (((long)ent * 2 + 15) & ~15L) + 32;
  const bool fast = n <= MAXN1 && need <= lds_bytes;
  const int lo = s1lo[b];
  if (threadIdx.x == 0) {
    hdr[(size_t)b * 4 + 0] = lo;
    hdr[(size_t)b * 4 + 1] = fast ? n : -1;
    hdr[(size_t)b * 4 + 2] = e0;
    hdr[(size_t)b * 4 + 3] = ent;
    if (!fast) atomicAdd(nslow, 1);
  }
  const int* list = tmp + (size_t)b * GSPX_TILE_TMPCAP;
  const int keep = n <= GSPX_TILE_TMPCAP ? n : 0;  // slow blocks beyond the cap keep no list
  for (int i = threadIdx.x; i < keep; i += 256) s1rows[lo + i] = list[i];
  for (int i = threadIdx.x; i < ent; i += 256) {
    const int c = col[e0 + i];
    int pos = 0;
    if (c < N && fast) {
      int a = 0, z = n;
      while (a < z) {
        const int mid = (a + z) >> 1;
        if (list[mid] < c) a = mid + 1; else z = mid;
      }
      pos = a;
    }
    lidx[e0 + i] = (u8)pos;  // fast blocks: pos < MAXN1 <= 256
  }
}
// kept rows per block for the scan (a slow block beyond the cap contributes none)
// most stored entries of a staged block
__global__ void k_tiles_entmax(const int* __restrict__ hdr, int nb, int* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < nb && hdr[(size_t)b * 4 + 1] >= 0) atomicMax(out, hdr[(size_t)b * 4 + 3]);
}

__global__ void k_tiles_keep(const int* __restrict__ n1, int nb, int* __restrict__ keep) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= nb) keep[b] = (b < nb && n1[b] <= GSPX_TILE_TMPCAP) ? n1[b] : 0;
}

}  // namespace gspx
