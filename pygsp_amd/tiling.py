"""Row tiles: one level for the LDS-staged recurrence step (k_step_tile; the device builds the same lists itself,
gspx_graph_build_gather_tiles - this module is its host-side model and the CPU tests' reference), two levels for
the kernels of the experimental build (k_newton_pair, k_cheb_pair in pygsp_amd/csrc/experimental).

One launch of that kernel applies TWO Horner steps

    g  = sA * (F h)[S1] + bA * h[S1] + gA * x[S1]          (on the block's 1-hop closure S1)
    h' = sB * (F g)[B]  + bB * g[B]  + gB * x[B]           (on the block's own rows B)

per workgroup, with h staged once in LDS on the 2-hop closure S2 and g kept in LDS, so per two
polynomial orders a row of the panel is read about once and written once (instead of 2 x 3
passes).  This module computes, from the engine's internal padded CSR, what the kernel needs:

    s1ptr[nb+1], s1rows[.]   rows of S1 per block (ascending; contains the block's own rows)
    s2ptr[nb+1], s2rows[.]   rows of S2 per block (ascending; S1 is a subset)
    lidx1[nnz_int]  uint16   for every stored entry of a block row: position of its column in S1
    occ_off[.], lidx2[.]     for every (block, row in S1) occurrence: where that row's entries'
                             positions in S2 start in lidx2 (entry order = the internal CSR order)

Pads (col == N) get index 0xFFFF.  It is a host-side (numpy) graph-setup step, like the vertex
ordering; nothing here touches signal data.
"""
import numpy as np

PAD = 0xFFFF


def build_tiles(rptr, rcol, N, block_rows=64, levels=2):
    """rptr: int32[N+1] internal row starts (low 2 bits may carry pad counts), rcol: int32[nnz_int]
    internal columns (pads == N).  Returns a dict of the arrays above plus max_n1 / max_n2.
    levels=1 stops after S1 / lidx1 (the LDS-staged recurrence step needs no more)."""
    rptr = (np.asarray(rptr, dtype=np.int64) & ~3)
    rcol = np.asarray(rcol, dtype=np.int64)
    BR = int(block_rows)
    nb = (N + BR - 1) // BR
    nnz = int(rptr[N])
    rowlen = np.diff(rptr)
    row_of_entry = np.repeat(np.arange(N, dtype=np.int64), rowlen)
    real = rcol[:nnz] < N
    M = N + 1

    # ---- level 1: S1(b) = distinct columns of the block's rows --------------------------------
    blk = row_of_entry // BR
    key1 = blk[real] * M + rcol[:nnz][real]
    s1key = np.unique(key1)
    s1blk = s1key // M
    s1rows = (s1key % M).astype(np.int32)
    s1ptr = np.searchsorted(s1blk, np.arange(nb + 1)).astype(np.int32)
    lidx1 = np.full(nnz, PAD, dtype=np.uint16)
    pos = np.searchsorted(s1key, key1)
    loc = pos - s1ptr[blk[real]]
    if loc.size and loc.max() >= PAD:
        raise ValueError("tile too large for 16-bit local indices")
    lidx1[real] = loc.astype(np.uint16)

    if levels == 1:
        n1 = np.diff(s1ptr)
        return {"block_rows": BR, "nb": nb, "s1ptr": s1ptr, "s1rows": s1rows, "lidx1": lidx1,
                "max_n1": int(n1.max()) if nb else 0, "mean_n1": float(n1.mean()) if nb else 0.0}

    # ---- level 2: every (block, row in S1) occurrence expands to that row's entries --------------
    occ_rows = s1rows.astype(np.int64)
    occ_len = rowlen[occ_rows]
    occ_off = np.zeros(occ_rows.size + 1, dtype=np.int64)
    np.cumsum(occ_len, out=occ_off[1:])
    total2 = int(occ_off[-1])
    # entry index (into rcol) of every expanded slot
    occ_of_slot = np.repeat(np.arange(occ_rows.size, dtype=np.int64), occ_len)
    within = np.arange(total2, dtype=np.int64) - occ_off[occ_of_slot]
    ent = rptr[occ_rows[occ_of_slot]] + within
    col2 = rcol[ent]
    real2 = col2 < N
    blk2 = s1blk[occ_of_slot]
    key2 = blk2[real2] * M + col2[real2]
    s2key = np.unique(key2)
    s2blk = s2key // M
    s2rows = (s2key % M).astype(np.int32)
    s2ptr = np.searchsorted(s2blk, np.arange(nb + 1)).astype(np.int32)
    lidx2 = np.full(total2, PAD, dtype=np.uint16)
    pos2 = np.searchsorted(s2key, key2)
    loc2 = pos2 - s2ptr[blk2[real2]]
    if loc2.size and loc2.max() >= PAD:
        raise ValueError("tile too large for 16-bit local indices")
    lidx2[real2] = loc2.astype(np.uint16)

    n1 = np.diff(s1ptr)
    n2 = np.diff(s2ptr)
    return {
        "block_rows": BR, "nb": nb,
        "s1ptr": s1ptr, "s1rows": s1rows, "s2ptr": s2ptr, "s2rows": s2rows,
        "lidx1": lidx1, "occ_off": occ_off.astype(np.uint32), "lidx2": lidx2,
        "max_n1": int(n1.max()) if nb else 0, "max_n2": int(n2.max()) if nb else 0,
        "mean_n1": float(n1.mean()) if nb else 0.0, "mean_n2": float(n2.mean()) if nb else 0.0,
    }


def emulate_pair(tiles, rptr, rcol, fval, N, h, x, A, B):
    """numpy model of k_newton_pair (same index arithmetic, block by block) - used by the CPU tests
    to validate the tile structure.  A = (s, b, g) of the first step, B of the second.
    Returns h' (N, ld)."""
    rp = (np.asarray(rptr, dtype=np.int64) & ~3)
    BR = tiles["block_rows"]
    out = np.zeros_like(h)
    for b in range(tiles["nb"]):
        s2 = tiles["s2rows"][tiles["s2ptr"][b]:tiles["s2ptr"][b + 1]]
        s1lo, s1hi = tiles["s1ptr"][b], tiles["s1ptr"][b + 1]
        s1 = tiles["s1rows"][s1lo:s1hi]
        tile_h = h[s2]                                   # phase 0
        tile_g = np.zeros((s1.size, h.shape[1]), dtype=h.dtype)
        for o, r in enumerate(s1):                       # phase 1
            lo = int(tiles["occ_off"][s1lo + o])
            s, e = rp[r], rp[r + 1]
            idx = tiles["lidx2"][lo:lo + (e - s)].astype(np.int64)
            val = fval[s:e]
            ok = idx != PAD
            acc = (val[ok, None] * tile_h[idx[ok]]).sum(axis=0)
            hself = tile_h[idx[0]]                       # entry 0 is the diagonal slot
            tile_g[o] = A[0] * acc + A[1] * hself + A[2] * x[r]
        for i in range(b * BR, min((b + 1) * BR, N)):    # phase 2
            s, e = rp[i], rp[i + 1]
            idx = tiles["lidx1"][s:e].astype(np.int64)
            val = fval[s:e]
            ok = idx != PAD
            acc = (val[ok, None] * tile_g[idx[ok]]).sum(axis=0)
            gself = tile_g[idx[0]]
            out[i] = B[0] * acc + B[1] * gself + B[2] * x[i]
    return out
