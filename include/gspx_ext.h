/*
 * gspx_ext.h - entry points of libgspx BESIDE the drop-in boundary of include/gspx.h: the rows SURVEY.md
 * section 8 marks "next" (operators on the same device CSR, nearest-neighbour / SBM graph construction),
 * the opt-in Newton-form evaluation, host-built tiles, the schedule export used by the CPU tests and the
 * bandwidth calibration kernels.  Same conventions as gspx.h (status codes, ownership, one ctx = one
 * stream).  A binding that only replaces pygsp.filters.approximations.cheby_op needs none of these.
 */
#ifndef GSPX_EXT_H
#define GSPX_EXT_H

#include "gspx.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* A polynomial of the scaled operator t = (2 / lmax) L - I evaluated as a PROGRAM of S steps on N x Nsig panels:
 *     h_0 = x;   h_{s+1} = scale_s * (2 t) h_s + beta_s * h_s + gamma_s * o_s;   y = h_S,
 * o_s = x for every step (old_is_x != 0: the Newton form above is such a program) or o_s = h_{s-1} (old_is_x == 0, gamma_0
 * ignored: the PRODUCT form - a real root r of the polynomial is one step with scale sigma / 2, beta -sigma r, gamma 0, which
 * reads one panel and writes one; a conjugate pair a +- ib is two steps, the second with gamma sigma^2 b^2).  The same
 * polynomial as approximations.py:93-112 evaluates when the program is built from its Chebyshev coefficients
 * (pygsp_amd.filters.cheb_to_product, behind filters.product_guard): 2.2 - 2.5 panel passes per order instead of 3 2/3.
 * Single filter, analysis.  _dev: device pointers; the other: host arrays.  S < 1 -> GSPX_ERR_COEFF. */
int gspx_poly_program_dev(gspx_graph* g, double lmax, int S, const double* scale, const double* beta, const double* gamma,
                          int old_is_x, int64_t Nsig, const void* x_dev, void* y_dev, double* kernel_ms);
int gspx_poly_program(gspx_graph* g, double lmax, int S, const double* scale, const double* beta, const double* gamma,
                      int old_is_x, int64_t Nsig, const void* x_host, void* y_host, double* kernel_ms);

/* The internal padded CSR pattern (engine vertex order; rowptr low 2 bits = pad counts, pads have col == N): what
 * host-built tiles (pygsp_amd/tiling.py) are computed from. */
int gspx_graph_download_internal(gspx_graph* g, int32_t* rowptr, int32_t* col);

/* Optional acceleration structure for gspx_cheby_filter* with ONE filter: one-level row tiles
 * (64-row blocks of the internal vertex order: per block the distinct rows it gathers, s1ptr /
 * s1rows; per stored entry the 16-bit position of its column in that list, lidx, pads = 0;
 * pygsp_amd/tiling.py builds them from gspx_graph_download_internal).  With tiles set (and option
 * "tile_gather" = 1, the default) every recurrence step stages the gathered panel in LDS
 * (k_step_tile).  block_rows == 0 drops the tiles.  stats (nullable): blocks, blocks on the
 * plain-gather path (tile too large for LDS), dynamic LDS bytes per workgroup. */
int gspx_graph_set_gather_tiles(gspx_graph* g, int block_rows, int nb, const int32_t* s1ptr,
                                const int32_t* s1rows, const uint16_t* lidx, int64_t* stats);

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
/* The differential operator of a DIRECTED graph, or of a graph with self-loops, from the caller's edge list
 * (Graph.get_edge_list, graph.py:1019-1029: every stored entry of W for a directed graph, the upper triangle
 * including the diagonal otherwise; sources non-decreasing).  The device graph was built from the symmetrised W
 * (graph.py:613-616), whose degrees are the reference's dw of the directed graph, so the D values of
 * difference.py:151-161 are formed on the device: -sqrt(w) / +sqrt(w), or -sqrt(w / dw[source]) /
 * +sqrt(w / dw[target]), divided by sqrt(2) when `directed`.  Replaces the edge list gspx_graph_n_edges /
 * gspx_graph_download_edges / gspx_grad_dev / gspx_div_dev work on (by default the upper triangle of the graph's
 * own Laplacian). */
int gspx_graph_set_edge_list(gspx_graph* g, int64_t n_edges, const int32_t* sources, const int32_t* targets,
                             const double* weights, int directed);

/* grad: y (n_edges x Nsig) = D^T x   (difference.py:168-244)
 * div:  z (N x Nsig)       = D y     (difference.py:246-331) */
int gspx_grad_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, void* y_dev, double* kernel_ms);
int gspx_div_dev(gspx_graph* g, int64_t Nsig, const void* y_dev, void* z_dev, double* kernel_ms);

/* ---- k-nearest-neighbour graph construction on the device ---------------
 * Replaces, for NNtype='knn' and 1..64 dimensions (a uniform grid in 1-3 dimensions; beyond that a tiled brute
 * force whose pair distances run on the matrix cores, candidates re-evaluated in the KD-tree's arithmetic),
 * the KD-tree query, the Gaussian weights and the symmetrisation of NNGraph
 * (pygsp/graphs/nngraphs/nngraph.py:213-226, 289-297):
 *   D, NN = KDTree(X).query(X, k + 1);  sigma = mean(D[:, 1:]);  w = exp(-D^2 / sigma);
 *   W = (W + W.T) / 2
 * coords: N x d doubles on the HOST, already centred / rescaled by the caller (nngraph.py:129-137).
 * sigma == 0 selects the mean neighbour distance.  metric: 0 euclidean, 1 manhattan, 2 max_dist (the
 * reference's dist_type; 'minkowski' with order 1, 2 or inf maps onto them).  symmetrize: 0 'average',
 * 1 'maximum' (= 'fill' for a k-NN matrix), 2 'tril', 3 'triu' (utils.symmetrize, utils.py:247-275).  Neighbours and distances
 * equal scipy's KD-tree bit for bit (ties ordered by vertex index); a point is never its own neighbour.  Context option "knn_f32": the candidate sweep beyond three
 * dimensions on the fp32 matrix cores - 1 (default) when its rounding margin is small against the bounds, 0 never,
 * 2 always; the selection is exact either way. */
typedef struct gspx_knn gspx_knn;
int gspx_knn_build(gspx_ctx* ctx, int64_t N, int d, const double* coords, int k, double sigma,
                   int metric, int symmetrize, gspx_knn** out);
int gspx_knn_destroy(gspx_knn* h);
int gspx_knn_info(gspx_knn* h, int64_t* nnz, double* sigma, double* build_ms);
/* how the neighbour search of the last build ran in more than three dimensions (tiled brute force, pair
 * distances on MFMA): out[0] sample size behind the per-query bounds, out[1] candidate capacity per query,
 * out[2] mean candidates per query, out[3] queries that took the exact scan (zeros for the 1-3-D grid search) */
int gspx_knn_search_stats(gspx_knn* h, double out[4]);
/* symmetric W as CSR (sorted columns), float64 */
int gspx_knn_download_w(gspx_knn* h, int32_t* indptr, int32_t* indices, double* data);
/* NN[:, 1:] and D[:, 1:] of the reference: N x k, nearest first (either may be NULL) */
int gspx_knn_download_neighbors(gspx_knn* h, int32_t* nn, double* dist);

/* Radius graphs: NNtype='radius' of NNGraph (nngraph.py:228-287) - neighbours within epsilon (the
 * KD-tree's ball query, squared distance <= epsilon^2), weights exp(-d^2 / sigma), sigma == 0 selects
 * the mean neighbour distance ("No neighbors found" -> GSPX_ERR_INVALID, as the reference's ValueError).
 * 1 to 64 dimensions: a grid of epsilon-sized cells up to 3-D; beyond that the candidate pairs come from an MFMA
 * distance sweep (a counting pass, then a filling pass into rows of exactly those lengths) and are tested in the
 * KD-tree's arithmetic.  Result read with gspx_knn_info / gspx_knn_download_w, freed with gspx_knn_destroy. */
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
/* The same sampler with the reference's other two switches (stochasticblockmodel.py:69-70, 128-130): flags bit 0
 * = directed (every ORDERED pair (r, c) is an entry W[r, c] = 1 with probability M[z_r][z_c]; M need not be
 * symmetric), bit 1 = self_loops (the pairs r == c take part; an undirected self-loop is one stored entry).
 * flags = 0 is gspx_sbm_build.  `connected=True` is a loop of the host layer over seeds (n_try). */
int gspx_sbm_build_ex(gspx_ctx* ctx, int64_t N, int k, const int32_t* order, const int64_t* bounds,
                      const double* M, uint64_t seed, int flags, gspx_knn** out);

/* Space-filling-curve keys of N points (coords: N x d doubles on the HOST, d >= 2; the first two /
 * three axes are used): curve 0 = Morton, 1 = Hilbert (2-D).  The engine's internal vertex order
 * for graphs with coordinates is the stable argsort of these keys (pygsp_amd.engine.locality_order). */
int gspx_curve_keys(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve, uint64_t* keys);

/* The vertex order itself: the stable argsort of those keys (what numpy.argsort(keys, kind="stable") returns),
 * by a radix sort on the device.  perm[new] = old, N int32 on the HOST. */
int gspx_curve_order(gspx_ctx* ctx, int64_t N, int d, const double* coords, int curve, int32_t* perm);

/* Graph set-up in ONE call, on the device: what pygsp/graphs/graph.py:98-176 (the checks of Graph.__init__),
 * Graph.is_directed (graph.py:357-405) and Graph.compute_laplacian (graph.py:510-630) do to W, plus this engine's
 * own preparation (CSR validation, curve order of the vertices and its locality score).  W (host CSR; data_dtype
 * GSPX_F32, GSPX_F64 or 2 = int64) is uploaded once and inspected in place; coords (N x d doubles on the HOST,
 * nullable) feed the internal order: order_mode 0 none, 1 auto (Hilbert in 2-D / Morton in 3-D, kept only when
 * it beats the graph's own order by 0.05 of locality score), 2 Morton, 3 Hilbert, 4 the permutation perm_in.
 * report[12]: [0] NaN entries, [1] infinite, [2] negative, [3] explicit zeros, [4] non-zero diagonal entries
 * (self-loops), [5] entries that differ from their mirror entry, a mirror that is not stored counting as zero
 * (> 0: W - W.T has a nonzero = Graph.is_directed), [6] CSR violations,
 * [7] 1 when an internal order is in use, [8] / [9] locality score x 1e9 of the graph's own / the curve order
 * (auto mode), [10] 0 = graph built, 1 = not built, [11] microseconds of the whole call.
 * NaN / inf / CSR violations: GSPX_ERR_INVALID with the reference's message.  A directed graph, or explicit zeros:
 * the Laplacian is built from (W + W.T) / 2 without stored zeros - utils.symmetrize(W, 'average'), graph.py:613-616,
 * utils.py:247-248; graph.py:126-128 - prepared on the device in the same call (transpose by radix sort, row merge;
 * degrees = its row sums = (in + out) / 2, graph.py:834-837); the caller keeps W itself (and drops its zeros).
 * Only a union pattern beyond 32-bit indices returns GSPX_OK with *out == NULL and report[10] == 1. */
int gspx_graph_setup(gspx_ctx* ctx, int64_t N, int64_t nnz, const int32_t* indptr, const int32_t* indices,
                     const void* data, int data_dtype, int lap_type, int compute_dtype, const double* coords,
                     int d, int order_mode, const int32_t* perm_in, int64_t report[12], gspx_graph** out);
/* The same set-up for the W a device builder left on the device (gspx_knn_build / gspx_radius_build /
 * gspx_sbm_build; the handle stays valid and still serves gspx_knn_download_w): the generator classes NNGraph /
 * Sensor / StochasticBlockModel / ErdosRenyi (nngraph.py:289-313, stochasticblockmodel.py:144-181 end in
 * Graph.__init__(W)) hand the builder's handle over - no download and re-upload of W; its host copy is made
 * when somebody reads G.W.  coords may be NULL (order_mode 0 or 4).  report / *out as gspx_graph_setup. */
int gspx_graph_setup_from_knn(gspx_knn* h, int lap_type, int compute_dtype, const double* coords, int d,
                              int order_mode, const int32_t* perm_in, int64_t report[12], gspx_graph** out);
/* The ingredients of Graph._get_upper_bound (graph.py:933-960: the smallest of four classical upper bounds of
 * lambda_max of the combinatorial Laplacian), taken in one pass while W was on the device - float64 graphs built
 * from W: out[0] max W_ij, out[1] max dw, out[2] max (dw_i + dw_j) over the stored entries, out[3] max (dw_i +
 * (W dw)_i / dw_i), NaN when a vertex has degree zero (numpy's 0 / 0; the reference's min() then ignores it).
 * The bounds are N out[0], 2 out[1], out[2], out[3]. */
int gspx_graph_lmax_bounds(gspx_graph* g, double out[4]);
/* the internal vertex order of a graph (perm[new] = old); GSPX_ERR_INVALID when it has none */
int gspx_graph_download_perm(gspx_graph* g, int32_t* perm);

/* Columns [j0, j0 + w) of the N x N identity as a row-major N x w panel in device memory (dtype GSPX_F32 /
 * GSPX_F64), queued on the context's stream: the input of Filter.compute_frame (filter.py:593-600 filters
 * np.identity(N)) produced where it is consumed. */
int gspx_identity_panel_dev(gspx_ctx* ctx, int dtype, int64_t N, int64_t j0, int64_t w, void* out_dev);

/* PCI address ("0000:c1:00.0", NUL-terminated) of HIP device `device`: what a host driver needs to find the NUMA
 * node the GPU hangs off (/sys/bus/pci/devices/<address>/numa_node) and pin the thread - and the packing threads
 * libgspx starts from it - that feeds this GPU to the cores next to it (pygsp_amd.multi). */
int gspx_device_pci_bus_id(int device, char* out, int capacity);

/* A signal cube between its two layouts, on the device (queued on the context's stream): the row-major
 * (N, S, F) tensor Filter.filter takes and returns (filter.py:146-328: vertices x signals x features) and the F
 * feature planes [f][n][s] the engine works on (the (Nf N, Nsig) stacking of approximations.py:88,
 * filter.py:315-316).  to_planes != 0: src is the cube, dst the planes; 0: the other way.  Needed only when a
 * device-resident array arrives in the other layout than the call reads it in (an (N, Nf) panel of Nf signals
 * taken as ONE signal with Nf features, filter.py:270-278). */
int gspx_planes_pack_dev(gspx_ctx* ctx, int dtype, int64_t N, int64_t S, int64_t F, const void* src_dev,
                         void* dst_dev, int to_planes);

/* Host-only: the step schedule the engine would run for (Nf, M) under the ctx's current
 * options, for CPU-side verification of the schedule logic (no device work).  Each of the
 * K = M-1 rows of `plan` is 4 + 3*Nf doubles:
 *   [scale, gamma, flush (0 none / 1 write / 2 accumulate), final (0/1),
 *    then for each filter f: w_new, w_cur, w_old]
 * `plan` must hold (M-1)*(4+3*Nf) doubles.  a1 = a2 = lmax/2 as approximations.py:93-96. */
int gspx_plan_describe(gspx_ctx* ctx, int Nf, int M, const double* coeffs, double* plan);
/* Host-only: the signal-column batches (widths[0 .. *n_batches), they add up to Nsig; 0 batches = the one-shot
 * form) and the host threads per direction the pipelined gspx_cheby_filter would use for a call of N x Nsig
 * elements of `dtype` with `planes_total` = Nf + 1 panels, under the options "host_pipeline" (mode),
 * "host_batch", "host_edge", "host_threads". */
int gspx_host_pipeline_describe(int mode, int64_t host_batch, int64_t host_edge, int64_t host_threads, int dtype,
                                int64_t N, int64_t Nsig, int planes_total, int64_t* widths, int capacity,
                                int* n_batches, int* threads);

/* Stage times of the LAST gspx_cheby_filter call on this context (milliseconds): out[0] wall time of the
 * pipelined call, out[1] packing (busiest host thread), out[2] host-to-device DMA (sum over batches), out[3]
 * kernels (sum of device times), out[4] device-to-host DMA, out[5] unpacking (busiest host thread), out[6]
 * batches (0: the call was not pipelined), out[7] signals per batch, out[8] host threads per direction. */
int gspx_last_host_timing(gspx_ctx* ctx, double out[9]);
/* Host clock of the last pipelined gspx_cheby_filter, per batch, in ms since the call began: [6 b + 0] packed,
 * [+1] H2D issued, [+2] kernels begun, [+3] kernels done, [+4] D2H done, [+5] unpacked.  *batches: how many there
 * were; out (capacity doubles) may be NULL. */
int gspx_last_host_timeline(gspx_ctx* ctx, double* out, int capacity, int* batches);

/* Calibration: read+write GB/s of the engine's 16-byte-per-lane streaming copy kernel over two
 * `bytes`-sized buffers (the measured HBM ceiling reported beside roofline fractions). */
int gspx_bench_copy(gspx_ctx* ctx, int64_t bytes, int iters, double* gbps);

/* Calibration: read-only GB/s of a `bytes`-sized buffer streamed `passes` times in one launch. */
int gspx_bench_read(gspx_ctx* ctx, int64_t bytes, int passes, double* gbps);

/* Placement tuning of a context's streamed workspaces for one single-filter analysis call (the arguments of
 * gspx_cheby_filter_dev with Nf = 1): `candidates` (1-32) physical backings are drawn - the previous ones held meanwhile,
 * so every draw gets other pages -, the call itself runs three times on each, the fastest stays in the context and
 * the others are released.  Why: on MI355X the recurrence on panels beyond the Infinity Cache runs 0.54-0.60 of 8 TB/s
 * depending on which physical pages back its work panels, and what a process draws first it keeps
 * (profiles/r06_placement.md).  out[i]: ms per recurrence launch with candidate i (0: the backing the context already
 * had, if any; 0: never drawn, memory ran out); out[candidates]: index kept.  stride_mb > 0: a pad of that many MB is
 * allocated and held before every further draw (released at the end), so that the candidates sample the card's memory at
 * that stride - fast and slow pages come in zones of tens of GB in allocation order.  y_dev holds the call's result
 * afterwards; results are bit-identical whichever backing is kept. */
int gspx_ctx_tune_placement(gspx_graph* g, double lmax, int M, const double* coeffs, int64_t Nsig, const void* x_dev,
                            void* y_dev, int candidates, int64_t stride_mb, double* out);

/* Calibration: total GB/s of n_read (0-4) read streams and n_write (0-2) write streams of bytes_per_stream each, walked
 * together by workgroups_per_cu persistent workgroups per CU, 16 bytes per lane (nt: bit 0 non-temporal loads, bit 1
 * non-temporal stores, bit 2 write stream w overwrites read stream w in place, bit 3 the recurrence step's walk - every
 * workgroup takes 32 KB blocks of its XCD's eighth of every stream - instead of a grid-stride sweep, bit 4 the streams lie
 * in the context's own T workspace instead of fresh allocations - GSPX_ERR_INVALID when it is smaller) - what the memory
 * system delivers to a read : write ratio with nothing else in the way. */
int gspx_bench_streams(gspx_ctx* ctx, int64_t bytes_per_stream, int n_read, int n_write, int nt, int workgroups_per_cu,
                       int iters, double* gbps);

/* Calibration, the MIX CEILING of the LDS-staged recurrence step: exactly the call
 * gspx_cheby_filter_dev(g, lmax, 1, M, coeffs, Nsig, x_dev, y_dev, GSPX_ANALYSIS, NULL) would make - the same plan, the
 * same M-1 k_step_tile launches on the same grid over the same buffers, the same LDS-DMA tile loads of the same row
 * lists, the same T_{k-2} / accumulator loads, entry stream, stores, flush steps, sweep directions and cache bits - with
 * the ARITHMETIC REMOVED from every launch: the per-entry LDS gathers and multiply-adds of the row products are
 * replaced by one tile read per row.  mode 1 keeps the two workgroup barriers of a pass, mode 2 drops them as well
 * (the access mix alone).  y_dev receives numbers without meaning.  Times are read with gspx_last_timing as after
 * any filter call.  Only for calls that run the wide builds (gather tiles, rows of more than 128 bytes); the one
 * launch per call that reads T_{k-2} from the caller's unpermuted panel (step 2, "fuse_input") runs unmodified.
 * What it answers: step time == mix time  =>  the step is bound by the memory system serving this access mix (on
 * this box), not by its arithmetic or LDS traffic; step time > mix time  =>  the compute phase is exposed. */
int gspx_bench_step_mix(gspx_graph* g, double lmax, int M, const double* coeffs, int64_t Nsig, const void* x_dev,
                        void* y_dev, int mode);

/* Calibration for graphs WITHOUT vertex locality (BASELINE configs 2, 3: Erdos-Renyi, block model): the rate
 * of random row gathers, measured with nothing else in the way.  n_gathers rows of row_bytes (64 / 128 / 256 /
 * 512; 16 bytes per lane, the step kernels' lane layout) are fetched from a panel of panel_rows rows by 32-bit
 * indices, in_flight (2 / 4 / 8 / 16) independent gathers per lane before the first use; no matrix values, no
 * FMA, no panel writes.  blocks == 1: indices uniform over the panel (ER); blocks > 1: the panel is cut into
 * that many row ranges and a gather lands in the range of the row it is issued "from" with probability p_intra
 * (SBM), each XCD walking a contiguous eighth of the stream; blocks == 0: UNIQUE rows - panel_rows a power of two,
 * n_gathers <= panel_rows, every row fetched at most once per launch in a scattered order (known bytes and no reuse:
 * the calibration of the fabric counters on gathers, tools/gather_calibration.py).  ms: per launch; gbps: row bytes
 * per second. */
int gspx_bench_gather(gspx_ctx* ctx, int64_t panel_rows, int row_bytes, int64_t n_gathers, int in_flight,
                      int blocks, double p_intra, int workgroups_per_cu, int iters, double* ms, double* gbps);

/* What RCCL itself reports, for self-validating multi-GPU records (bench.py --gpus N): out[0] RCCL version code
 * (ncclGetVersion, e.g. 22203; 0 when RCCL cannot be loaded), out[1] ranks the communicator spans as the communicator
 * says (ncclCommCount) - for comm == NULL the device set of this process's last RCCL gspx_gather (0: none ran) -,
 * out[2] this handle's rank (ncclCommUserRank; -1 for NULL). */
int gspx_comm_info(gspx_comm* comm, int64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif /* GSPX_EXT_H */
