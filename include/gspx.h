/*
 * gspx.h — C-ABI of libgspx: MI355X (gfx950) Chebyshev graph-filtering engine.
 *
 * Drop-in boundary for ONE hot path of epfl-lts2/pygsp (a pure-Python library, so the
 * "FFI" a maintainer binds is ctypes):
 *
 *   pygsp/filters/approximations.py:58-114   cheby_op(G, c, signal)      -> gspx_cheby_filter*
 *   pygsp/filters/filter.py:303-322          Filter.filter analysis/synthesis dispatch
 *   pygsp/graphs/graph.py:510-630            Graph.compute_laplacian     -> gspx_graph_create_from_w
 *   pygsp/graphs/graph.py:830-838            Graph.dw                    -> gspx_graph_download_dw
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; gspx_last_error() returns a
 *     thread-local message for the last failing call on this thread.  Nothing throws across
 *     the boundary, nothing calls exit().
 *   - host pointers are borrowed for the duration of the call only.
 *   - device memory is owned by handles (ctx / graph / buf) and freed by *_destroy / *_free.
 *   - one ctx = one device + one HIP stream.  Calls on one ctx must be serialised by the
 *     caller; different ctx may be driven from different threads (ctypes drops the GIL).
 *   - no torch / numpy types appear here: plain pointers and sizes.
 *
 * This header is the boundary of the path itself.  The entry points beside it - operators on the same
 * device CSR, graph construction, the opt-in Newton evaluation, host-built tiles, calibration kernels -
 * are declared in gspx_ext.h.
 */
#ifndef GSPX_H
#define GSPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gspx_ctx gspx_ctx;
typedef struct gspx_graph gspx_graph;
typedef struct gspx_buf gspx_buf;

/* element types */
#define GSPX_F32 0
#define GSPX_F64 1

/* Laplacian types (graph.py:618-628) */
#define GSPX_LAP_COMBINATORIAL 0
#define GSPX_LAP_NORMALIZED 1

/* filter modes (filter.py:307-322) */
#define GSPX_ANALYSIS 0  /* x: [N][Nsig]       -> y: [Nf][N][Nsig] */
#define GSPX_SYNTHESIS 1 /* x: [Nf][N][Nsig]   -> y: [N][Nsig]     */

/* status codes */
#define GSPX_OK 0
#define GSPX_ERR_INVALID 1 /* bad argument (shape / dtype / null)   -> ValueError  */
#define GSPX_ERR_COEFF 2   /* M < 2 (approximations.py:83-84)       -> TypeError   */
#define GSPX_ERR_HIP 3     /* HIP runtime failure                   -> RuntimeError*/
#define GSPX_ERR_NODEVICE 4
#define GSPX_ERR_OOM 5     /* device allocation failed, nothing written yet -> RuntimeError (safe to retry) */

const char* gspx_last_error(void);
const char* gspx_version(void);

/* ---- devices / contexts --------------------------------------------------------------- */
int gspx_device_count(int* n);
int gspx_ctx_create(int device, gspx_ctx** out);
int gspx_ctx_destroy(gspx_ctx* ctx);
int gspx_ctx_sync(gspx_ctx* ctx);
/* Integer options of a context.  Unknown key -> GSPX_ERR_INVALID.  The defaults are what the engine
 * is measured with; the others exist so that every kernel variant stays testable against the oracle.
 *   "kernel"        plain gather kernels (graphs without gather tiles): 0 auto; 1 lane-group panel kernel
 *                   with scalar metadata; 2 narrow (sub-wave rows, 1-4 signals); 5 LDS-staged CSR slice
 *   "tile_gather"   1 (default): recurrence steps stage the gathered panel in LDS when the graph carries
 *                   gather tiles (k_step_tile); 0: plain gather kernels
 *   "fuse_input"    1 (default): with gather tiles, steps 1-2 of a single-filter call read the caller's
 *                   panel in place instead of copying it into the internal vertex order first
 *   "combine"       0 auto; 1 fused flush (accumulate every third step); 2 deferred (keep all T_k, one
 *                   combine pass; the default for filterbanks)
 *   "synthesis"     0 (default) one vector-coefficient Clenshaw recurrence (K sparse products for any
 *                   Nf); 1 the reference's per-filter loop (K*Nf products, filter.py:317-321)
 *   "graph_launch"  2 (default) an analysis call that repeats the previous one exactly is recorded as a
 *                   hipGraph and replayed when its panel is at most 32 MB (launch-bound); 1 always;
 *                   0 never.  A replayed call reports one total time (gspx_last_timing out[0] == out[1])
 *   "ws_limit_mb" / "max_batch"   workspace budget per call / cap on signals per batch
 *   "alternate_sweep" 1 (default) odd steps sweep the rows from the end (Infinity-Cache reuse)
 *   "xcd_remap"     1 (default) contiguous row ranges per XCD in the plain gather kernels
 *   "tile_nt"       k_step_tile non-temporal accesses, bit 0 matrix entries, bit 1 accumulator, bit 2
 *                   T_{k-2} rows, bit 3 T_k stores; -1 (default): 5 for panels of 192 MiB and more
 *   "tile_workgroups"   persistent workgroups of k_step_tile (0: 2 per CU; what fits for the small builds)
 *   "tile_pad"      1 (default) panels whose rows are not made of 16-byte pieces take k_step_tile with padded
 *                   rows (a single signal only on graphs beyond the L2s); 2 always; 0 never.  "tile_min_row"
 *                   (16) narrowest rows in bytes k_step_tile takes; "tile_lg" 2 / 4 / 8: no build narrower than that
 *   "vec", "rows_per_wave", "narrow_g_log2", "waves_per_block"   launch shapes of the plain gather
 *                   kernels (0 / -1 = auto)
 *   "edge_vertex_walk" 1 (default) grad / div walk the vertices in the internal order; 0 edge order
 *   "host_pipeline" gspx_cheby_filter: 1 (default) large calls pipelined in column batches, 2 always, 0 never;
 *                   "host_batch" signals per batch, "host_edge" of the first / last one, "host_threads" (0 = auto)
 *   "streamed_alloc" 1 (default) workspaces from scrambled 2 MB chunks (+2..8 % bandwidth; a retired range
 *                   keeps its address space: read-only "retired_va_mb"); 0 plain hipMalloc, the safe mode
 *                   (also GSPX_STREAMED_ALLOC=0 in the environment) */
int gspx_ctx_set_option(gspx_ctx* ctx, const char* key, int64_t value);
int gspx_ctx_get_option(gspx_ctx* ctx, const char* key, int64_t* value);

/* ---- device buffers (for device-resident chaining and honest kernel-only timing) ------- */
int gspx_buf_alloc(gspx_ctx* ctx, int64_t bytes, gspx_buf** out);
int gspx_buf_free(gspx_buf* buf);
int gspx_buf_upload(gspx_buf* buf, const void* host, int64_t bytes);
int gspx_buf_download(gspx_buf* buf, void* host, int64_t bytes);
int gspx_buf_ptr(gspx_buf* buf, void** device_ptr); /* raw device pointer (interop) */
int gspx_buf_bytes(gspx_buf* buf, int64_t* bytes);

/* ---- the path's one collective: gather of the ranks' outputs to a root, RCCL over xGMI ----------------
 * (the reference has no multi-device path, nothing is replaced).  The recurrence
 * never communicates: independent graphs / signal columns per GPU.  Every block travels as one grouped
 * ncclSend / ncclRecv pair, so each peer uses its own xGMI link.  RCCL is loaded on first use (dlopen).
 *
 * Single-process form (one process driving several contexts / GPUs from threads): concatenates
 * parts[0..n) - each a buffer of its own context - into root_out, in argument order; returns when all
 * have landed.  ctxs may be null (or ctxs[i] == the context parts[i] was allocated on).  Option
 * "gather_rccl" of the ROOT's context: 1 (default) RCCL between devices, peer copies (hipMemcpyPeerAsync
 * on the source streams) if RCCL is unavailable; 0 peer copies only; 2 every block through RCCL, also
 * same-device ones (self send / recv). */
int gspx_gather(gspx_ctx** ctxs, int n, gspx_buf** parts, gspx_buf* root_out);
/* One-process-per-GPU form.  The launcher (torch.distributed.run, mpirun, ...) only moves the 128-byte id
 * from rank 0 to the other ranks; everything else is RCCL inside the library.
 *   gspx_comm_available  1 when RCCL could be loaded
 *   gspx_comm_unique_id  rank 0: a fresh id (ncclGetUniqueId)
 *   gspx_comm_create     every rank, collectively: the communicator of `nranks` ranks on ctx's device
 *   gspx_comm_gather     every rank, collectively, with the same bytes[nranks] table: rank r's block
 *                        part_dev (bytes[r] bytes, device memory) lands at offset sum(bytes[0..r)) of
 *                        root_out_dev on rank `root` (NULL elsewhere).  Queued on the context's stream,
 *                        behind the filter that produced the block; returns when this rank's transfers are
 *                        complete.  ms (nullable): device time of the exchange on this rank. */
typedef struct gspx_comm gspx_comm;
#define GSPX_COMM_ID_BYTES 128
int gspx_comm_available(void);
int gspx_comm_unique_id(unsigned char id[GSPX_COMM_ID_BYTES]);
int gspx_comm_create(gspx_ctx* ctx, int nranks, int rank, const unsigned char id[GSPX_COMM_ID_BYTES],
                     gspx_comm** out);
int gspx_comm_destroy(gspx_comm* comm);
int gspx_comm_gather(gspx_comm* comm, const void* part_dev, const int64_t* bytes, int root,
                     void* root_out_dev, double* ms);

/* ---- graphs ---------------------------------------------------------------------------- */
/* Upload the (symmetric, canonical CSR: sorted indices, no duplicates, no explicit zeros)
 * weight matrix W and build the Laplacian ON DEVICE.  Replaces graph.py:618-628 (+ dw,
 * graph.py:830-838).  Directed graphs must be symmetrised by the caller first
 * (graph.py:613-616).  data_dtype: dtype of `data`; compute_dtype: dtype of L and of all
 * filtering arithmetic.  `perm` (nullable, length N): vertex ordering used INTERNALLY only
 * (perm[new] = old); inputs and outputs of every call stay in the caller's vertex order. */
int gspx_graph_create_from_w(gspx_ctx* ctx, int64_t N, int64_t nnz, const int32_t* indptr,
                             const int32_t* indices, const void* data, int data_dtype,
                             int lap_type, int compute_dtype, const int32_t* perm,
                             gspx_graph** out);
/* Upload a host-built Laplacian (bit-parity mode; also for directed graphs whose L the
 * reference builds through utils.symmetrize).  Same CSR requirements. */
int gspx_graph_create_from_l(gspx_ctx* ctx, int64_t N, int64_t nnz, const int32_t* indptr,
                             const int32_t* indices, const void* data, int data_dtype,
                             int compute_dtype, const int32_t* perm, gspx_graph** out);
int gspx_graph_destroy(gspx_graph* g);
int gspx_graph_n(gspx_graph* g, int64_t* N);
/* nnz of the canonical Laplacian (explicit zeros dropped, as scipy does) */
int gspx_graph_nnz_l(gspx_graph* g, int64_t* nnz);
/* number of stored entries of the internal padded layout (diagnostics / roofline maths) */
int gspx_graph_nnz_internal(gspx_graph* g, int64_t* nnz);
/* canonical L in the caller's vertex order; data in the graph's compute dtype */
int gspx_graph_download_l(gspx_graph* g, int32_t* indptr, int32_t* indices, void* data);
/* weighted degree (only for graphs created from W); compute dtype */
int gspx_graph_download_dw(gspx_graph* g, void* dw);
/* wall-clock milliseconds of the device-side build (H2D excluded) */
int gspx_graph_build_ms(gspx_graph* g, double* ms);

/* ---- the hot path ---------------------------------------------------------------------- */
/* Order-(M-1) Chebyshev filtering, whole K-loop inside one call.
 *   coeffs : host, Nf x M row-major float64 (as compute_cheby_coeff returns them; c[.,0] is
 *            halved inside, approximations.py:103)
 *   x, y   : DEVICE pointers, compute dtype, row-major, contiguous (layouts above)
 *   lmax   : the value the reference would read from G.lmax (approximations.py:93)
 *   kernel_ms (nullable): device time of the whole call (HIP events on the ctx stream)
 * Errors: M < 2 -> GSPX_ERR_COEFF; Nf < 1, Nsig < 0, lmax <= 0, null pointers -> GSPX_ERR_INVALID.
 * Nsig == 0 is a no-op. */
int gspx_cheby_filter_dev(gspx_graph* g, double lmax, int Nf, int M, const double* coeffs,
                          int64_t Nsig, const void* x_dev, void* y_dev, int mode,
                          double* kernel_ms);
/* Same with HOST pointers (pageable memory is fine): what a caller of pygsp/filters/filter.py:146-328 hands
 * over.  Large calls are cut into signal-column batches and pipelined over pinned staging buffers - host
 * threads pack batch b+2, DMA of b+1, kernels of b, DMA of b-1 and unpacking of b-2 overlap - so the call
 * costs about max(PCIe, kernels) instead of their sum; small calls are one copy in, the kernels, one copy
 * out (option "host_pipeline").  On a graph with gather tiles every batch runs the kernel family of the one-shot
 * call: identical bytes (without tiles, batches under 32-byte rows are equal to rounding only).
 * kernel_ms: device time of the kernels alone. */
int gspx_cheby_filter(gspx_graph* g, double lmax, int Nf, int M, const double* coeffs,
                      int64_t Nsig, const void* x_host, void* y_host, int mode,
                      double* kernel_ms);

/* Gather tiles of the default recurrence kernel (k_step_tile): per 64-row block of the internal vertex
 * order the distinct rows it gathers, per stored entry the 8-bit position of its column in that list, built
 * on the device (per-block sort / unique in LDS).  Built by the host layer for graphs of >= 32k vertices and
 * kept when the order is local.  stats (nullable, 4 values): blocks, blocks on the plain-gather path, LDS
 * bytes per workgroup, total rows in the gather lists.  (Host-built tiles: gspx_ext.h.) */
int gspx_graph_build_gather_tiles(gspx_graph* g, int64_t* stats);

/* timing breakdown of the LAST filter call on this graph's ctx (milliseconds, HIP events):
 *   out[0] total device time, out[1] time inside the recurrence-step launches only,
 *   out[2] number of step launches, out[3] permute-in/copy time, out[4] combine time */
int gspx_last_timing(gspx_ctx* ctx, double out[5]);

/* Largest eigenvalue of L by Lanczos on the device (replaces the ARPACK call of
 * pygsp/graphs/graph.py:911-917).  Deterministic (fixed start vector).  Returns the largest Ritz
 * value (<= lambda_max); stops when the residual of the Ritz pair is below `tol` * value (an
 * eigenvalue of L lies within that distance; the reference asks ARPACK for 5e-3) or after
 * `max_iter` steps; *converged (nullable) tells which: 0 = the step budget ran out first, and the value
 * may sit well below lambda_max - the reference raises ValueError in that case (ArpackNoConvergence,
 * graph.py:918-919) and so must the caller.  The caller applies the reference's 1 % margin (graph.py:920). */
int gspx_lanczos_lmax(gspx_graph* g, int max_iter, double tol, double* lmax, int* iterations,
                      int* converged);

#ifdef __cplusplus
}
#endif
#endif /* GSPX_H */
