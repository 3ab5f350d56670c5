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

const char* gspx_last_error(void);
const char* gspx_version(void);

/* ---- devices / contexts --------------------------------------------------------------- */
int gspx_device_count(int* n);
int gspx_ctx_create(int device, gspx_ctx** out);
int gspx_ctx_destroy(gspx_ctx* ctx);
int gspx_ctx_sync(gspx_ctx* ctx);
/* Integer options of a context.  Unknown key -> GSPX_ERR_INVALID.  The defaults are what the engine
 * is measured with; the others exist so that every kernel variant stays testable against the oracle.
 *   "kernel"        0 auto; 1 lane-group panel kernel; 2 narrow (sub-wave rows, 1-4 signals);
 *                   3 / 4 wave-row kernels; 5 LDS-staged CSR slice.  Ignored when gather tiles apply
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
 *   "tile_workgroups" / "pair_workgroups"   persistent workgroups of k_step_tile / k_newton_pair (0: 2 per CU)
 *   "vec", "rows_per_wave", "narrow_g_log2", "waves_per_block"   launch shapes of the plain gather
 *                   kernels (0 / -1 = auto)
 *   "newton_pair"   1 (default) Newton-form filtering runs two orders per launch when the graph carries
 *                   pair tiles (gspx_graph_set_tiles)
 *   "edge_vertex_walk" 1 (default) grad / div walk the vertices in the internal order; 0 edge order */
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
/* Same with HOST pointers (one H2D + one D2H around the device call). */
int gspx_cheby_filter(gspx_graph* g, double lmax, int Nf, int M, const double* coeffs,
                      int64_t Nsig, const void* x_host, void* y_host, int mode,
                      double* kernel_ms);

/* The same polynomial in NEWTON form (single filter, analysis), evaluated by Horner:
 *     y = sum_{j=0..K} d_j prod_{i<j} (Lt - r_i I) x,    Lt = (L - a2 I)/a1,  a1 = a2 = lmax/2
 * `nodes` = r_0..r_{K-1}, `dcoef` = d_0..d_K (host, float64).  A two-term recurrence: 3 panel
 * passes per order and no accumulator, against 3 + 2/3 for the three-term Chebyshev recurrence
 * of gspx_cheby_filter*.  The caller derives (nodes, dcoef) from the reference's Chebyshev
 * coefficients in exact arithmetic (pygsp_amd/filters.py::cheb_to_newton), so both entry points
 * evaluate the identical polynomial; they agree to rounding (~1e-14 in float64).
 * x: [N][Nsig], y: [N][Nsig].  K < 1 -> GSPX_ERR_COEFF. */
int gspx_newton_filter_dev(gspx_graph* g, double lmax, int K, const double* nodes,
                           const double* dcoef, int64_t Nsig, const void* x_dev, void* y_dev,
                           double* kernel_ms);
int gspx_newton_filter(gspx_graph* g, double lmax, int K, const double* nodes, const double* dcoef,
                       int64_t Nsig, const void* x_host, void* y_host, double* kernel_ms);

/* Optional acceleration structure for gspx_newton_filter*: two-level row tiles (32-row blocks) of
 * the internal vertex order, computed on the host from the internal pattern
 * (pygsp_amd/tiling.py).  With tiles set (and option "newton_pair" = 1, the default) two Horner
 * steps run per launch with the panel staged in LDS: the pass moves fewer bytes than the
 * algorithmic count of two steps.  block_rows == 0 drops the tiles. */
int gspx_graph_download_internal(gspx_graph* g, int32_t* rowptr, int32_t* col);
int gspx_graph_set_tiles(gspx_graph* g, int block_rows, int nb, const int32_t* s1ptr,
                         const int32_t* s1rows, const int32_t* s2ptr, const int32_t* s2rows,
                         const uint16_t* lidx1, const uint32_t* occ_off, int64_t n_lidx2,
                         const uint16_t* lidx2, int max_n1, int max_n2);
/* out[0] row blocks, out[1] blocks handled by the unstaged fallback kernel (tiles too large to hold
 * their matrix entries in LDS, or rows longer than 32 entries), out[2] dynamic LDS bytes per
 * workgroup of that fallback kernel (the staged kernel always takes 80 KB), out[3] rows per block
 * (0: no tiles set) */
int gspx_graph_tile_stats(gspx_graph* g, int64_t out[4]);

/* Optional acceleration structure for gspx_cheby_filter* with ONE filter: one-level row tiles
 * (64-row blocks of the internal vertex order: per block the distinct rows it gathers, s1ptr /
 * s1rows; per stored entry the 16-bit position of its column in that list, lidx, pads = 0;
 * pygsp_amd/tiling.py builds them from gspx_graph_download_internal).  With tiles set (and option
 * "tile_gather" = 1, the default) every recurrence step stages the gathered panel in LDS
 * (k_step_tile).  block_rows == 0 drops the tiles.  stats (nullable): blocks, blocks on the
 * plain-gather path (tile too large for LDS), dynamic LDS bytes per workgroup. */
int gspx_graph_set_gather_tiles(gspx_graph* g, int block_rows, int nb, const int32_t* s1ptr,
                                const int32_t* s1rows, const uint16_t* lidx, int64_t* stats);

/* The same tiles computed on the device from the internal CSR (per-block sort / unique in LDS).
 * stats (nullable, 4 values): blocks, blocks on the plain-gather path, LDS bytes per workgroup,
 * total rows in the gather lists. */
int gspx_graph_build_gather_tiles(gspx_graph* g, int64_t* stats);

/* ---- operators on the same device CSR --------------------------------
 * Panels are DEVICE pointers (gspx_buf_ptr or any other device allocation), row-major N x Nsig,
 * compute dtype of the graph, caller's vertex order.
 *
 * y = L x: the product inside Graph.dirichlet_energy (pygsp/graphs/graph.py:702) and inside the
 * operator of learning.regression_tikhonov (pygsp/learning.py:330). */
int gspx_laplacian_apply_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, void* y_dev,
                             double* kernel_ms);
/* gram_host[Nsig*Nsig] (double, row-major, HOST) = X^T (L X): Graph.dirichlet_energy,
 * graph.py:642-702 (a scalar for one signal). */
int gspx_dirichlet_energy_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, double* gram_host,
                              double* kernel_ms);
/* Tikhonov regression with tau > 0 (pygsp/learning.py:324-337): solves (diag(M) + tau L) x = M y,
 * one conjugate-gradient run per column with scipy.sparse.linalg.cg's recurrence and stopping rule
 * (x0 = 0, ||r|| < max(atol, rtol ||b||); scipy's defaults are rtol 1e-5, atol 0, maxiter 10 N).
 * mask_dev: N values of the compute dtype (1 = measured, 0 = not).  iterations: Nsig ints (HOST)
 * or NULL. */
int gspx_tikhonov_cg_dev(gspx_graph* g, double tau, const void* mask_dev, int64_t Nsig,
                         const void* y_dev, void* x_dev, double rtol, double atol, int64_t maxiter,
                         int32_t* iterations, double* kernel_ms);
/* Differential operator D (L = D D^T) of an UNDIRECTED graph without self loops created from W
 * (pygsp/graphs/difference.py:26-166).  Edges = stored entries (i, j > i) in row-major order, the
 * order of Graph.get_edge_list (graph.py:1019-1029).  Built on the device at first use.
 * download: any output may be NULL; d_source / d_target are D[i, k] at the edge's source (negative)
 * and target (positive). */
int gspx_graph_n_edges(gspx_graph* g, int64_t* n_edges);
int gspx_graph_download_edges(gspx_graph* g, int32_t* sources, int32_t* targets, void* weights,
                              void* d_source, void* d_target);
/* grad: y (n_edges x Nsig) = D^T x   (difference.py:168-244)
 * div:  z (N x Nsig)       = D y     (difference.py:246-331) */
int gspx_grad_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, void* y_dev, double* kernel_ms);
int gspx_div_dev(gspx_graph* g, int64_t Nsig, const void* y_dev, void* z_dev, double* kernel_ms);

/* ---- k-nearest-neighbour graph construction on the device ---------------
 * Replaces, for NNtype='knn', dist_type='euclidean', symmetrize_type='average' and 1..3 dimensions,
 * the KD-tree query, the Gaussian weights and the symmetrisation of NNGraph
 * (pygsp/graphs/nngraphs/nngraph.py:213-226, 289-297):
 *   D, NN = KDTree(X).query(X, k + 1);  sigma = mean(D[:, 1:]);  w = exp(-D^2 / sigma);
 *   W = (W + W.T) / 2
 * coords: N x d doubles on the HOST, already centred / rescaled by the caller (nngraph.py:129-137).
 * sigma == 0 selects the mean neighbour distance.  metric: 0 euclidean, 1 manhattan, 2 max_dist (the
 * reference's dist_type; 'minkowski' with order 1, 2 or inf maps onto them).  symmetrize: 0 'average',
 * 1 'maximum' (= 'fill' for a k-NN matrix), 2 'tril', 3 'triu' (utils.symmetrize, utils.py:247-275).  Neighbours and distances
 * equal scipy's KD-tree bit for bit (ties ordered by vertex index); a point is never its own neighbour. */
typedef struct gspx_knn gspx_knn;
int gspx_knn_build(gspx_ctx* ctx, int64_t N, int d, const double* coords, int k, double sigma,
                   int metric, int symmetrize, gspx_knn** out);
int gspx_knn_destroy(gspx_knn* h);
int gspx_knn_info(gspx_knn* h, int64_t* nnz, double* sigma, double* build_ms);
/* symmetric W as CSR (sorted columns), float64 */
int gspx_knn_download_w(gspx_knn* h, int32_t* indptr, int32_t* indices, double* data);
/* NN[:, 1:] and D[:, 1:] of the reference: N x k, nearest first (either may be NULL) */
int gspx_knn_download_neighbors(gspx_knn* h, int32_t* nn, double* dist);

/* Radius graphs: NNtype='radius' of NNGraph (nngraph.py:228-287) - neighbours within epsilon (the
 * KD-tree's ball query, squared distance <= epsilon^2), weights exp(-d^2 / sigma), sigma == 0 selects
 * the mean neighbour distance ("No neighbors found" -> GSPX_ERR_INVALID, as the reference's ValueError).
 * Result read with gspx_knn_info / gspx_knn_download_w, freed with gspx_knn_destroy. */
int gspx_radius_build(gspx_ctx* ctx, int64_t N, int d, const double* coords, double epsilon,
                      double sigma, int metric, gspx_knn** out);

/* Stochastic block model / Erdos-Renyi graph sampled on the device: every unordered pair (r, c) of
 * distinct vertices is an edge (unit weight) independently with probability M[z_r][z_c] - the
 * distribution of pygsp/graphs/stochasticblockmodel.py:125-144 (directed=False, self_loops=False)
 * and erdosrenyi.py (k = 1), in O(edges) instead of the reference's N^2 Python loop.  The random
 * stream is the engine's own (counter-based), so graphs equal the reference's in distribution, not
 * bit for bit.  order: the vertices grouped by block (a stable argsort of z), bounds[k + 1]: where
 * each block starts in it, M: k x k symmetric, row-major.  The result is read with
 * gspx_knn_info / gspx_knn_download_w and freed with gspx_knn_destroy. */
int gspx_sbm_build(gspx_ctx* ctx, int64_t N, int k, const int32_t* order, const int64_t* bounds,
                   const double* M, uint64_t seed, gspx_knn** out);

/* Space-filling-curve keys of N points (coords: N x d doubles on the HOST, d >= 2; the first two /
 * three axes are used): curve 0 = Morton, 1 = Hilbert (2-D).  The engine's internal vertex order
 * for graphs with coordinates is the stable argsort of these keys (pygsp_amd.engine.locality_order). */
int gspx_curve_keys(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve, uint64_t* keys);

/* Columns [j0, j0 + w) of the N x N identity as a row-major N x w panel in device memory (dtype GSPX_F32 /
 * GSPX_F64), queued on the context's stream: the input of Filter.compute_frame (filter.py:593-600 filters
 * np.identity(N)) produced where it is consumed. */
int gspx_identity_panel_dev(gspx_ctx* ctx, int dtype, int64_t N, int64_t j0, int64_t w, void* out_dev);

/* timing breakdown of the LAST filter call on this graph's ctx (milliseconds, HIP events):
 *   out[0] total device time, out[1] time inside the recurrence-step launches only,
 *   out[2] number of step launches, out[3] permute-in/copy time, out[4] combine time */
int gspx_last_timing(gspx_ctx* ctx, double out[5]);

/* Host-only: the step schedule the engine would run for (Nf, M) under the ctx's current
 * options, for CPU-side verification of the schedule logic (no device work).  Each of the
 * K = M-1 rows of `plan` is 4 + 3*Nf doubles:
 *   [scale, gamma, flush (0 none / 1 write / 2 accumulate), final (0/1),
 *    then for each filter f: w_new, w_cur, w_old]
 * `plan` must hold (M-1)*(4+3*Nf) doubles.  a1 = a2 = lmax/2 as approximations.py:93-96. */
int gspx_plan_describe(gspx_ctx* ctx, int Nf, int M, const double* coeffs, double* plan);

/* Largest eigenvalue of L by Lanczos on the device (replaces the ARPACK call of
 * pygsp/graphs/graph.py:911-917).  Deterministic (fixed start vector).  Returns the largest Ritz
 * value (<= lambda_max); stops when the residual of the Ritz pair is below `tol` * value (an
 * eigenvalue of L lies within that distance; the reference asks ARPACK for 5e-3) or after
 * `max_iter` steps; *converged (nullable) tells which: 0 = the step budget ran out first, and the value
 * may sit well below lambda_max - the reference raises ValueError in that case (ArpackNoConvergence,
 * graph.py:918-919) and so must the caller.  The caller applies the reference's 1 % margin (graph.py:920). */
int gspx_lanczos_lmax(gspx_graph* g, int max_iter, double tol, double* lmax, int* iterations,
                      int* converged);

/* Calibration: read+write GB/s of the engine's 16-byte-per-lane streaming copy kernel over two
 * `bytes`-sized buffers (the measured HBM ceiling reported beside roofline fractions). */
int gspx_bench_copy(gspx_ctx* ctx, int64_t bytes, int iters, double* gbps);

/* Calibration: read-only GB/s of a `bytes`-sized buffer streamed `passes` times in one launch. */
int gspx_bench_read(gspx_ctx* ctx, int64_t bytes, int passes, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* GSPX_H */
