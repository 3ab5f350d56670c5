// experimental/gspx_newton_pair.hip.h - the fused two-step Newton kernel (rounds 1-2).  Compiled only with
// -DGSPX_EXPERIMENTAL (make EXTRA=-DGSPX_EXPERIMENTAL): the default library evaluates the Newton form with
// single k_step_tile steps, which measured faster.
#pragma once

namespace gspx {

// ---------------------------------------------------------------------------------------------
// FUSED NEWTON PAIR: two Horner steps of the Newton-form evaluation in ONE pass over the panel.
//     g  = sA (F h)[r] + bA h[r] + gA x[r]      for r in S1 = 1-hop closure of the block's rows
//     h' = sB (F g)[i] + bB g[i] + gB x[i]      for the block's own rows i
// h is staged once in LDS on the 2-hop closure S2, g lives only in LDS: per TWO polynomial orders
// a panel row is read ~once (plus halo) and written once, where two separate steps read it twice
// as a gather source, stream x twice and write twice.  This is the temporally blocked form the
// HBM-bound analysis in DESIGN.md asks for: it moves fewer bytes than the algorithmic count.
// Tiles (S1, S2, 16-bit local indices) come from pygsp_amd/tiling.py.
//   workgroup = 512 threads = 32 groups of 16 lanes x 16 bytes (one 256-byte row chunk per group);
//   blockIdx.y walks the 256-byte column chunks of wider panels.
// ---------------------------------------------------------------------------------------------
#define GSPX_PAD16 0xFFFFu
typedef u16 u16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct PairArgs {
  const int* rowptr;
  const T* fval;
  const int* hdr;      // [nb][8]: s1lo, n1, s2lo, n2, occ_base, ent1, rp0, ent2 (fallback kernel)
  const int* hdr_s;    // [nsb][8], stageable blocks only: s1lo, n1 | n2 << 16, s2lo, block, then as hdr
  const int4* desc;    // per (block, S1 row): internal row, first entry in fval, lidx2 offset | entries << 24
                       // (staged kernel; 0 when that does not fit), lidx2 offset (fallback kernel)
  const int* s2rows;
  const u16* lidx1;
  const u16* lidx2;
  const T* h_in;
  const T* x;
  T* h_out;
  int N;
  u32 ld;
  u32 panel_bytes, fval_bytes, lidx1_bytes, lidx2_bytes;
  int lds_bytes;  // dynamic LDS of the staged kernel: [bottom: h tile, S1 entries | top: g tile, own entries]
  int top_bytes;  // size of the top part
  int nb;      // row blocks
  int ncol;    // 256-byte column chunks per row
  int nsb;     // stageable blocks (entries of hdr_s)
  int per_xcd; // stageable blocks per XCD
  T sA, bA, gA, sB, bB, gB;
  int final;
  T* y;
  u32 ldy;
  const int* perm;
};

// sum_j val_j * tile[idx_j] over one padded CSR row, two 4-entry chunks per trip (their index and
// value loads are issued together); tile rows are 16 lanes wide.  Also returns the row's first
// gathered value (entry 0 = the diagonal slot = the row itself).  PADS: indices may be 0xFFFF.
template <typename T, typename V, bool PADS>
__device__ __forceinline__ V tile_row_dot(const T* __restrict__ val, const u16* __restrict__ idx, int len,
                                          const V* tile, int lane16, V& self) {
  typedef T T4 __attribute__((ext_vector_type(4)));
  V acc = 0;
  self = 0;
  for (int j = 0; j < len; j += 8) {
    const bool two = j + 4 < len;
    u16x4 ia = *(const u16x4*)(idx + j);
    const T4 va = *(const T4*)(val + j);
    u16x4 ib = *(const u16x4*)(idx + (two ? j + 4 : j));
    T4 vb = *(const T4*)(val + (two ? j + 4 : j));
    if (!two) vb = 0;
    if (PADS) {
      ia.x = ia.x == GSPX_PAD16 ? 0 : ia.x; ia.y = ia.y == GSPX_PAD16 ? 0 : ia.y;
      ia.z = ia.z == GSPX_PAD16 ? 0 : ia.z; ia.w = ia.w == GSPX_PAD16 ? 0 : ia.w;
      ib.x = ib.x == GSPX_PAD16 ? 0 : ib.x; ib.y = ib.y == GSPX_PAD16 ? 0 : ib.y;
      ib.z = ib.z == GSPX_PAD16 ? 0 : ib.z; ib.w = ib.w == GSPX_PAD16 ? 0 : ib.w;
    }
    const V t0 = tile[ia.x * 16 + lane16];
    const V t1 = tile[ia.y * 16 + lane16];
    const V t2 = tile[ia.z * 16 + lane16];
    const V t3 = tile[ia.w * 16 + lane16];
    const V t4 = tile[ib.x * 16 + lane16];
    const V t5 = tile[ib.y * 16 + lane16];
    const V t6 = tile[ib.z * 16 + lane16];
    const V t7 = tile[ib.w * 16 + lane16];
    if (j == 0) self = t0;
    acc += va.x * t0;  // pad values are 0
    acc += va.y * t1;
    acc += va.z * t2;
    acc += va.w * t3;
    acc += vb.x * t4;
    acc += vb.y * t5;
    acc += vb.z * t6;
    acc += vb.w * t7;
  }
  return acc;
}

// Fused Newton pair, persistent workgroups.  A workgroup (512 threads = 32 groups of 16 lanes)
// walks a contiguous run of stageable 32-row blocks (hdr_s lists them); workgroups are dealt
// round-robin to the 8 XCDs, so workgroup w = (xcd = w % 8, j = w / 8) takes the j-th slice of the
// xcd-th eighth of the list: each XCD's L2 sees one contiguous run of the internal order and the
// halo rows a block stages are its neighbours' own rows.  Per block and 256-byte column chunk:
//   phase 0  h on the 2-hop closure S2 -> LDS; gA * x on S1 -> LDS (g tile); for the first chunk
//            also the matrix entries (value + 16-bit tile position) of the S1 rows and of the
//            block's rows -> LDS, one entry per lane.  All global loads of the pass are issued
//            here; the row lists / descriptors they depend on were prefetched one block ahead
//            and the block headers two blocks ahead, so one memory latency is exposed per block.
//   phase 1  g = sA F h + bA h + gA x on S1, from LDS into LDS
//   phase 2  h' = sB F g + bB g + gB x on the block's rows, from LDS; the only store of the pass
template <typename T, int NCOL>  // NCOL = 1: one column chunk per row (straight-line passes); 0: a.ncol chunks
__global__ __launch_bounds__(512, 4) void k_newton_pair(const PairArgs<T> a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef typename VT<T, VEC>::t V;
  extern __shared__ __attribute__((aligned(16))) unsigned char gspx_smem[];

  const int tid = threadIdx.x;
  const int lane16 = tid & 15;
  const int grp = tid >> 4;  // 0..31
  // XCD x owns the stageable blocks [x * per_xcd, (x + 1) * per_xcd) of hdr_s; its nwx workgroups
  // take them round-robin, so at any moment they work on nwx consecutive blocks and find each
  // other's rows in the XCD's L2.  A workgroup does all column chunks of a block back to back:
  // the matrix entries are staged once per block.
  const int nwx = (int)(gridDim.x >> 3);
  const int xlo = (int)(blockIdx.x & 7) * a.per_xcd;
  int k1 = xlo + a.per_xcd;
  if (k1 > a.nsb) k1 = a.nsb;
  const int k0 = xlo + (int)(blockIdx.x >> 3);
  if (k0 >= k1) return;

  // 32-bit offsets into buffer descriptors; an offset with bit 31 set is out of range and loads 0
  constexpr u32 POISON = 0x80000000u;
  const rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_in, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.panel_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)a.fval, 0, a.fval_bytes, 0x00020000);
  const rsrc_t ri1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.lidx1, 0, a.lidx1_bytes, 0x00020000);
  const rsrc_t ri2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.lidx2, 0, a.lidx2_bytes, 0x00020000);
  const u32 ldb = a.ld * (u32)sizeof(T);

  struct Hdr { int4 p, q; };   // p: s1lo, n1 | n2 << 16, s2lo, block; q: occ_base, ent1, rp0, ent2
  struct Desc { int x, y, z; };  // row, first entry in fval, lidx2 offset | entries << 24
  struct Meta { Desc d[4]; int rows[8]; int rs, re; };
  auto load_hdr = [&](int k) {
    Hdr h;
    h.p = *(const int4*)(a.hdr_s + (size_t)k * 8);
    h.q = *(const int4*)(a.hdr_s + (size_t)k * 8 + 4);
    return h;
  };
  auto load_meta = [&](const Hdr& h) {
    Meta m;
    const int n1 = h.p.y & 0xFFFF, n2 = h.p.y >> 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = grp + 32 * t;
      const int4 dd = a.desc[h.p.x + (o < n1 ? o : 0)];
      m.d[t].x = dd.x;
      m.d[t].y = dd.y;
      m.d[t].z = dd.z;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int u = grp + 32 * t;
      m.rows[t] = a.s2rows[h.p.z + (u < n2 ? u : n2 - 1)];
    }
    int r = h.p.w * 32 + grp;
    r = r < a.N ? r : 0;
    m.rs = a.rowptr[r];
    m.re = a.rowptr[r + 1];
    return m;
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int wave = tid >> 6;
  // LDS (see PairArgs): bottom = h tile + entries of the S1 rows (dead after phase 1),
  //                     top    = g tile + entries of the block's rows (live through phase 2)
  unsigned char* const top = gspx_smem + (a.lds_bytes - a.top_bytes);
  V* const tile_h = (V*)gspx_smem;
  V* const tile_g = (V*)top;
  // h on the S2 rows of (M, n2), column chunk at byte offset cb -> h tile, straight into LDS:
  // the 4 groups of a wave stage 4 consecutive tile rows = 1 KB contiguous, 16 bytes per lane
  auto stage_h = [&](const Meta& m, int n2, u32 cb) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (grp + 32 * t < n2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr)(gspx_smem + (wave * 4 + 32 * t) * 256), 16,
                                                 (u32)m.rows[t] * ldb + cb, 0, 0, 0);
    }
  };
  auto chunk_off = [&](int c) {
    const u32 col0 = (c * 16 + lane16) * VEC;
    return col0 < a.ld ? col0 * (u32)sizeof(T) : POISON;
  };

  // headers are uniform: they travel in VGPRs while in flight and move to SGPRs only when due
  auto uniform = [](Hdr h) {
    Hdr u;
    u.p.x = __builtin_amdgcn_readfirstlane(h.p.x); u.p.y = __builtin_amdgcn_readfirstlane(h.p.y);
    u.p.z = __builtin_amdgcn_readfirstlane(h.p.z); u.p.w = __builtin_amdgcn_readfirstlane(h.p.w);
    u.q.x = __builtin_amdgcn_readfirstlane(h.q.x); u.q.y = __builtin_amdgcn_readfirstlane(h.q.y);
    u.q.z = __builtin_amdgcn_readfirstlane(h.q.z); u.q.w = __builtin_amdgcn_readfirstlane(h.q.w);
    return u;
  };
  Hdr H = uniform(load_hdr(k0));
  Meta M = load_meta(H);
  Hdr Hn = uniform(load_hdr(k0 + nwx < k1 ? k0 + nwx : k0));
  stage_h(M, H.p.y >> 16, chunk_off(0));
  // lanes with nothing to stage write here (end of the bottom part, never read); one slot per
  // lane of the wave, or the writes would pile up on one LDS bank
  const int dump_v = a.lds_bytes - a.top_bytes - 1024 + (tid & 63) * (int)sizeof(T);
  const int dump_i = a.lds_bytes - a.top_bytes - 512 + (tid & 63) * 2;

  int k = k0;
  // one pass = one column chunk c of block k; returns false after the workgroup's last pass
  auto pass = [&](const int c, const bool first, const bool last) __attribute__((always_inline)) {
    const int n1 = H.p.y & 0xFFFF, n2 = H.p.y >> 16, b = H.p.w;
    const u32 occ_base = (u32)H.q.x;
    const int ent1 = H.q.y, rp0 = H.q.z, ent2 = H.q.w;
    T* const mval1 = (T*)(tile_h + n2 * 16);
    u16* const midx1 = (u16*)(mval1 + ent1);
    T* const mval2 = (T*)(tile_g + n1 * 16);
    u16* const midx2 = (u16*)(mval2 + ent2);
    const int row2 = b * 32 + grp;  // one own row per group
    const bool ok2 = row2 < a.N;
    const int row2c = ok2 ? row2 : 0;
    const int s2s = M.rs & ~3;
    const int len2 = (M.re & ~3) - s2s;
    const int m2 = s2s - rp0;
    const u32 col0 = (c * 16 + lane16) * VEC;
    const bool on = col0 < a.ld;
    const u32 cb = on ? col0 * (u32)sizeof(T) : POISON;

    // entries of this group's S1 rows and own row: lane l takes entries l and l + 16
    u16 ei[5][2];
    T ev[5][2];
    if (first) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int e = lane16 + 16 * q;
          const u32 pm = e < (M.d[t].z >> 24) ? 0u : POISON;
          ei[t][q] = (u16)__builtin_amdgcn_raw_buffer_load_b16(ri2, ((occ_base + (u32)(M.d[t].z & 0xFFFFFF) + e) * 2u) | pm, 0, 0);
          ev[t][q] = VT<T, 1>::bload(rv, ((u32)(M.d[t].y + e) * (u32)sizeof(T)) | pm);
        }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = lane16 + 16 * q;
        const u32 pm = e < len2 ? 0u : POISON;
        ei[4][q] = (u16)__builtin_amdgcn_raw_buffer_load_b16(ri1, ((u32)(s2s + e) * 2u) | pm, 0, 0);
        ev[4][q] = VT<T, 1>::bload(rv, ((u32)(s2s + e) * (u32)sizeof(T)) | pm);
      }
    }
    V x1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x1[t] = VT<T, VEC>::bload(rx, (u32)M.d[t].x * ldb + cb);
    int dz[4], dw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dz[t] = M.d[t].z >> 24;
      dw[t] = M.d[t].z & 0xFFFFFF;
    }
    // last chunk of the block: the next block's row lists and the header after that go in flight
    Hdr Hv;
    if (last) {
      M = load_meta(Hn);
      Hv = load_hdr(k + 2 * nwx < k1 ? k + 2 * nwx : k);
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the pass is issued before the first wait

    // S1 entries -> bottom (nobody reads them between phase 1 of the previous block and here); a
    // lane with nothing to stage writes to its dump slot: no divergent branch for a load to sink into
    if (first) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int e = lane16 + 16 * q;
          const bool ok = grp + 32 * t < n1 && e < dz[t];
          const int av = ok ? (int)((unsigned char*)(mval1 + dw[t] + e) - gspx_smem) : dump_v;
          const int ai = ok ? (int)((unsigned char*)(midx1 + dw[t] + e) - gspx_smem) : dump_i;
          *(T*)(gspx_smem + av) = ev[t][q];
          *(u16*)(gspx_smem + ai) = ei[t][q] == GSPX_PAD16 ? (u16)0 : ei[t][q];
        }
    }
    // every wave waits for its own tile loads: __syncthreads() lowers to `s_waitcnt lgkmcnt(0); s_barrier`
    // and does not wait for `buffer_load ... lds` (vmcnt); see k_step_tile
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    __syncthreads();  // h tile and S1 entries in place; everybody is done with the previous phase 2
    // the header is the youngest load issued above and the entries (older) have landed: this wait is
    // short, and no wait of the next pass will have to reach past the tile loads issued further down
    Hdr Hnn = Hn;
    if (last) Hnn = uniform(Hv);
    const V x2 = VT<T, VEC>::bload(rx, (u32)row2c * ldb + cb);  // due at the end of phase 2
    if (first) {      // own-row entries -> top (read in phase 2)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = lane16 + 16 * q;
        const bool ok = ok2 && e < len2;
        const int av = ok ? (int)((unsigned char*)(mval2 + m2 + e) - gspx_smem) : dump_v;
        const int ai = ok ? (int)((unsigned char*)(midx2 + m2 + e) - gspx_smem) : dump_i;
        *(T*)(gspx_smem + av) = ev[4][q];
        *(u16*)(gspx_smem + ai) = ei[4][q] == GSPX_PAD16 ? (u16)0 : ei[4][q];
      }
    }
    // ---- phase 1 ----------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = grp + 32 * t;
      if (o < n1) {
        V self;
        const V acc = lds_row_dot<T, V>(mval1 + dw[t], midx1 + dw[t], dz[t], tile_h, lane16, self);
        tile_g[o * 16 + lane16] = a.sA * acc + a.bA * self + a.gA * x1[t];
      }
    }
    __syncthreads();  // g tile (and the own-row entries) complete; the h tile is free
    bool more = true;
    if (last) {
      H = Hn;
      Hn = Hnn;
      more = k + nwx < k1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) stage_h(M, H.p.y >> 16, chunk_off(last ? 0 : c + 1));
    // ---- phase 2 ----------------------------------------------------------------------------
    if (ok2) {
      V self;
      const V acc = lds_row_dot<T, V>(mval2 + m2, midx2 + m2, len2, tile_g, lane16, self);
      const V hn = a.sB * acc + a.bB * self + a.gB * x2;
      if (on) {
        if (a.final) {
          const size_t orow = a.perm ? (size_t)a.perm[row2] : (size_t)row2;
          *(V*)(a.y + orow * a.ldy + col0) = hn;
        } else {
          *(V*)(a.h_out + (size_t)row2 * a.ld + col0) = hn;
        }
      }
    }
    if (last) k += nwx;
    return more;
  };
  for (;;) {
    if constexpr (NCOL == 1) {
      if (!pass(0, true, true)) break;
    } else {
      bool more = true;
      for (int c = 0; c < a.ncol; ++c) more = pass(c, c == 0, c == a.ncol - 1);
      if (!more) break;
    }
  }
}

// The same pass for the blocks k_newton_pair leaves out (tiles too large to stage the entries
// next to them, rows longer than 32): entries read from global memory.  blocks = list of them.
template <typename T>
__global__ __launch_bounds__(512) void k_newton_pair_g(const PairArgs<T> a, const int* __restrict__ blocks) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef typename VT<T, VEC>::t V;
  extern __shared__ __attribute__((aligned(16))) unsigned char gspx_smem[];
  const int tid = threadIdx.x;
  const int lane16 = tid & 15;
  const int grp = tid >> 4;
  const int v = blockIdx.x;
  const int b = blocks[v / a.ncol];
  const u32 col0 = ((v % a.ncol) * 16 + lane16) * VEC;
  const bool on = col0 < a.ld;
  const int4 h0 = *(const int4*)(a.hdr + (size_t)b * 8);
  const int s1lo = h0.x, n1 = h0.y, s2lo = h0.z, n2 = h0.w;
  const u32 occ_base = (u32)a.hdr[(size_t)b * 8 + 4];
  V* tile_h = (V*)gspx_smem;
  V* tile_g = tile_h + n2 * 16;
  for (int u0 = grp; u0 < n2; u0 += 32 * 8) {
    int rows[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int u = u0 + 32 * t;
      rows[t] = a.s2rows[s2lo + (u < n2 ? u : n2 - 1)];
    }
    V hv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      hv[t] = 0;
      if (on && u0 + 32 * t < n2) hv[t] = *(const V*)(a.h_in + (size_t)rows[t] * a.ld + col0);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (u0 + 32 * t < n2) tile_h[(u0 + 32 * t) * 16 + lane16] = hv[t];
  }
  __syncthreads();
  for (int o = grp; o < n1; o += 32) {
    const int4 dd = a.desc[s1lo + o];
    V xr = 0;
    if (on) xr = *(const V*)(a.x + (size_t)dd.x * a.ld + col0);
    V self;
    const int dlen = (a.rowptr[dd.x + 1] & ~3) - dd.y;
    const V acc = tile_row_dot<T, V, true>(a.fval + dd.y, a.lidx2 + occ_base + (u32)dd.w, dlen, tile_h, lane16, self);
    tile_g[o * 16 + lane16] = a.sA * acc + a.bA * self + a.gA * xr;
  }
  __syncthreads();
  const int row2 = b * 32 + grp;
  if (row2 < a.N) {
    const int s2s = a.rowptr[row2] & ~3;
    const int len2 = (a.rowptr[row2 + 1] & ~3) - s2s;
    V x2 = 0;
    if (on) x2 = *(const V*)(a.x + (size_t)row2 * a.ld + col0);
    V self;
    const V acc = tile_row_dot<T, V, true>(a.fval + s2s, a.lidx1 + s2s, len2, tile_g, lane16, self);
    const V hn = a.sB * acc + a.bB * self + a.gB * x2;
    if (on) {
      if (a.final) {
        const size_t orow = a.perm ? (size_t)a.perm[row2] : (size_t)row2;
        *(V*)(a.y + orow * a.ldy + col0) = hn;
      } else {
        *(V*)(a.h_out + (size_t)row2 * a.ld + col0) = hn;
      }
    }
  }
}

}  // namespace gspx
