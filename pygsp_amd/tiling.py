"""Row tiles of the LDS-staged recurrence step (k_step_tile).  The device builds the same lists itself
(gspx_graph_build_gather_tiles); this module is their host-side (numpy) model, the reference of the CPU tests and the
source of `DeviceGraph.enable_gather_tiles`.

From the engine's internal padded CSR it computes, per 64-row block of the internal vertex order,

    s1ptr[nb+1], s1rows[.]   the distinct rows the block's entries touch (S1: ascending, contains the block's rows)
    lidx1[nnz_int]  uint16   for every stored entry of a block row: position of its column in S1

Pads (col == N) get index 0xFFFF.  It is a graph-setup step, like the vertex ordering; nothing here touches signal
data.  (The two-level tiles of the retired two-orders-per-launch kernels went with them in round 6:
profiles/r05_pair_experiment.md.)
"""
import numpy as np

PAD = 0xFFFF


def build_tiles(rptr, rcol, N, block_rows=64):
    """rptr: int32[N+1] internal row starts (low 2 bits may carry pad counts), rcol: int32[nnz_int]
    internal columns (pads == N).  Returns a dict of the arrays above plus max_n1 / mean_n1."""
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

    n1 = np.diff(s1ptr)
    return {"block_rows": BR, "nb": nb, "s1ptr": s1ptr, "s1rows": s1rows, "lidx1": lidx1,
            "max_n1": int(n1.max()) if nb else 0, "mean_n1": float(n1.mean()) if nb else 0.0}
