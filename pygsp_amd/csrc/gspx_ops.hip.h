// gspx_ops.hip.h - host side of the operators that reuse the engine's CSR.
// Included at the end of gspx.hip (uses its helpers).  All entry points take DEVICE pointers to
// row-major N x Nsig panels of the graph's compute dtype, in the caller's vertex order.
#pragma once

#include "gspx_ops_kernels.hip.h"

// byte offsets col*ld*sizeof(T) for this panel width (LDS / wave-row kernels), cached on the graph

// out = scale * (vals . cur) + beta * cur on internal-order panels (one launch of the step kernel);
// with y != null the result goes to y in the caller's order instead (rows y[perm[i]])
template <typename T>
static int spmm_internal(gspx_graph* g, const T* vals, T scale, T beta, const T* cur, T* out,
                         unsigned ld, T* y, unsigned ldy, const int* s1rows = nullptr) {
  gspx_ctx* ctx = g->ctx;
  Options opt = ctx->opt;
  if (tile_usable<T>(g, opt, ld, y ? y : out, y ? ldy : ld)) {  // LDS-staged gather, as the filter steps
    TileArgs<T> t{};
    t.s1rows = s1rows;  // non-null: cur is a panel in the caller's vertex order (lists mapped through perm)
    t.cur = cur;
    t.old = cur;
    t.out = out;
    t.racc = const_cast<T*>(cur);  // unused (flush == 0)
    t.y = y;
    t.ldy = ldy;
    t.perm = g->has_perm ? g->perm.as<int>() : nullptr;
    t.scale = scale;
    t.gamma = T(0);
    t.beta = beta;
    t.flush = 0;
    t.final = y ? 1 : 0;
    return launch_step_tile<T>(g, opt, t, ld, ctx->stream, vals);
  }
  int veccap = 4;
  if (y)
    while (veccap > 1 && ((ldy % veccap) != 0 || (((uintptr_t)y / sizeof(T)) % veccap) != 0))
      veccap /= 2;
  const Shape shape = choose_shape(opt, sizeof(T), ld, veccap);
  CHK(prepare_coff<T>(g, shape, ld, ctx->stream));
  StepArgs<T> a{};
  a.rowptr = g->rptr.as<int>();
  a.col = g->rcol.as<int>();
  a.val = vals;
  a.N = (int)g->N;
  a.ld = ld;
  a.curbytes = (u32)((size_t)g->N * ld * sizeof(T));
  a.cur = cur;
  a.old = cur;
  a.out = out;
  a.scale = scale;
  a.gamma = T(0);
  a.beta = beta;
  a.final = y ? 1 : 0;
  a.y = y;
  a.ldy = ldy;
  a.perm = g->has_perm ? g->perm.as<int>() : nullptr;
  launch_step<T>(a, shape, opt, ctx->stream, g->coff.as<unsigned>());
  return GSPX_OK;
}

static int64_t ops_max_ld(gspx_graph* g, size_t elt, int panels) {
  const Options& opt = g->ctx->opt;
  const size_t rowb = (size_t)g->N * elt;
  int64_t max_ld = (int64_t)((((size_t)1 << 31) - 65536) / std::max<size_t>(rowb, 1));
  const size_t budget = (size_t)std::max<int64_t>(opt.ws_limit_mb, 1) << 20;
  max_ld = std::min<int64_t>(max_ld, std::max<int64_t>(1, (int64_t)(budget / (rowb * panels))));
  if (opt.max_batch > 0) max_ld = std::min<int64_t>(max_ld, opt.max_batch);
  if (max_ld >= 4) max_ld &= ~(int64_t)3;
  return max_ld;
}

template <typename T> static int permute_panel(gspx_graph* g, const T* x, unsigned ldx, T* out,
                                               unsigned ld, const int* perm) {
  int pvec = 4;
  if (sizeof(T) == 8) pvec = 2;
  while (pvec > 1 && ((ldx % pvec) != 0 || (ld % pvec) != 0 ||
                      (((uintptr_t)x / sizeof(T)) % pvec) != 0 ||
                      (((uintptr_t)out / sizeof(T)) % pvec) != 0))
    pvec /= 2;
  launch_permute_in<T>(x, ldx, out, ld, (int)g->N, perm, pvec, g->ctx->stream);
  return GSPX_OK;
}

// ---- y = L x -------------------------------------------------------------------------------------
template <typename T>
static int lap_apply_t(gspx_graph* g, int64_t Nsig, const T* x, T* y, double* ms) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int64_t N = g->N;
  if (ms) *ms = 0;
  if (N == 0 || Nsig == 0) return GSPX_OK;
  const int64_t max_ld = ops_max_ld(g, sizeof(T), 1);
  if (max_ld < 1) return set_err(GSPX_ERR_INVALID, "graph too large: one signal column exceeds 2 GiB");
  const int* perm = g->has_perm ? g->perm.as<int>() : nullptr;
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  // one batch, every block on the LDS path: the step kernel gathers straight from x (the tile lists mapped
  // through the vertex order) and writes y in the caller's order - one launch, no copies
  const unsigned char *xb = (const unsigned char*)x, *yb = (const unsigned char*)y;
  const size_t pb = (size_t)N * Nsig * sizeof(T);
  if (Nsig <= max_ld && ctx->opt.fuse_input && g->gt_slow == 0 && ((uintptr_t)x % 16) == 0 &&
      (xb + pb <= yb || yb + pb <= xb) && (!g->has_perm || g->gt_ns1 > 0) &&
      tile_usable<T>(g, ctx->opt, (unsigned)Nsig, y, (unsigned)Nsig)) {
    CHK(ensure_s1nat(g, st));
    CHK(spmm_internal<T>(g, g->rval.as<T>(), T(1), T(0), x, nullptr, (unsigned)Nsig, y, (unsigned)Nsig,
                         g->has_perm ? g->gt_s1nat.as<int>() : nullptr));
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    float f1 = 0;
    HIPCHK(hipEventElapsedTime(&f1, ctx->ev[0], ctx->ev[1]));
    if (ms) *ms = f1;
    return GSPX_OK;
  }
  for (int64_t c0 = 0; c0 < Nsig; c0 += max_ld) {
    const unsigned ld = (unsigned)std::min<int64_t>(max_ld, Nsig - c0);
    CHK(ctx->ws_t.ensure((size_t)N * ld * sizeof(T) + 256));
    T* P0 = ctx->ws_t.as<T>();
    CHK(permute_panel<T>(g, x + c0, (unsigned)Nsig, P0, ld, perm));
    // the internal values array rval IS L (k_factor derives F from it)
    CHK(spmm_internal<T>(g, g->rval.as<T>(), T(1), T(0), P0, nullptr, ld, y + c0, (unsigned)Nsig));
  }
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, ctx->ev[0], ctx->ev[1]));
  if (ms) *ms = f;
  return GSPX_OK;
}

extern "C" int gspx_laplacian_apply_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, void* y_dev,
                                        double* kernel_ms) {
  if (g) replay_reset(g->ctx);
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  if (Nsig < 0) return set_err(GSPX_ERR_INVALID, "negative number of signals");
  if (Nsig > 0 && g->N > 0 && (!x_dev || !y_dev)) return set_err(GSPX_ERR_INVALID, "null signal pointer");
  HIPCHK(hipSetDevice(g->ctx->device));
  return g->dtype == GSPX_F32
             ? lap_apply_t<float>(g, Nsig, (const float*)x_dev, (float*)y_dev, kernel_ms)
             : lap_apply_t<double>(g, Nsig, (const double*)x_dev, (double*)y_dev, kernel_ms);
}

// ---- Dirichlet energy: gram = X^T (L X), Nsig x Nsig, double, row-major, on the host -----------------
template <typename T>
static int dirichlet_t(gspx_graph* g, int64_t Nsig, const T* x, double* gram, double* ms) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int64_t N = g->N;
  if (ms) *ms = 0;
  for (int64_t i = 0; i < Nsig * Nsig; ++i) gram[i] = 0.0;
  if (N == 0 || Nsig == 0) return GSPX_OK;
  int64_t max_ld = ops_max_ld(g, sizeof(T), 2);
  if (max_ld < Nsig)
    return set_err(GSPX_ERR_INVALID, "dirichlet_energy: %lld signals do not fit the workspace "
                   "(raise ws_limit_mb)", (long long)Nsig);
  const unsigned ld = (unsigned)Nsig;
  CHK(ctx->ws_t.ensure((size_t)2 * N * ld * sizeof(T) + 256));
  T* P0 = ctx->ws_t.as<T>();
  T* P1 = P0 + (size_t)N * ld;
  // per-wave partial Gram matrices, reduced on the device (at most 512 MB of them)
  const int64_t nw_cap = std::max<int64_t>(4, ((int64_t)512 << 20) / ((int64_t)ld * ld * 8));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(512, nw_cap / 4), (N + 255) / 256));
  const int nwaves = nb * 4;
  DevMem partial, gsum;
  CHK(partial.alloc((size_t)nwaves * ld * ld * sizeof(double)));
  CHK(gsum.alloc((size_t)ld * ld * sizeof(double)));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  CHK(permute_panel<T>(g, x, ld, P0, ld, g->has_perm ? g->perm.as<int>() : nullptr));
  CHK(spmm_internal<T>(g, g->rval.as<T>(), T(1), T(0), P0, P1, ld, nullptr, 0));
  // the sums do not depend on the vertex order: both panels stay in the internal order
  for (int a0 = 0; a0 < (int)ld; a0 += 64)
    for (int c0 = 0; c0 < (int)ld; c0 += 64) {
      const int na = std::min(64, (int)ld - a0), nc = std::min(64, (int)ld - c0);
      hipLaunchKernelGGL((k_gram_mfma<T>), dim3(nb), dim3(256), 0, st, P0, P1, (int)N, (int)ld, a0, na, c0,
                         nc, partial.as<double>());
    }
  hipLaunchKernelGGL(k_colsum, dim3(ld * ld), dim3(64), 0, st, partial.as<double>(), nwaves, (int)(ld * ld),
                     gsum.as<double>());
  HIPCHK(hipMemcpyAsync(gram, gsum.p, (size_t)ld * ld * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, ctx->ev[0], ctx->ev[1]));
  if (ms) *ms = f;
  return GSPX_OK;
}

extern "C" int gspx_dirichlet_energy_dev(gspx_graph* g, int64_t Nsig, const void* x_dev,
                                         double* gram_host, double* kernel_ms) {
  if (g) replay_reset(g->ctx);
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  if (Nsig < 0 || Nsig > 2048) return set_err(GSPX_ERR_INVALID, "dirichlet_energy: 0 <= Nsig <= 2048");
  if (Nsig > 0 && (!gram_host || (g->N > 0 && !x_dev))) return set_err(GSPX_ERR_INVALID, "null pointer");
  HIPCHK(hipSetDevice(g->ctx->device));
  return g->dtype == GSPX_F32 ? dirichlet_t<float>(g, Nsig, (const float*)x_dev, gram_host, kernel_ms)
                              : dirichlet_t<double>(g, Nsig, (const double*)x_dev, gram_host, kernel_ms);
}

// ---- Tikhonov regression, tau > 0: (diag(M) + tau L) x = M y by conjugate gradients ----------------
// One independent system per column, all columns advanced together; the recurrence and the stopping
// rule are those of scipy.sparse.linalg.cg (x0 = 0, no preconditioner; stop when
// ||r|| < max(atol, rtol ||b||), checked at the top of every iteration), which is what
// learning.regression_tikhonov calls column by column (learning.py:324-337).
template <typename T>
static int tikhonov_t(gspx_graph* g, double tau, const T* mask, int64_t Nsig, const T* y, T* x,
                      double rtol, double atol, int64_t maxiter, int32_t* iters, double* ms) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int64_t N = g->N;
  if (ms) *ms = 0;
  if (N == 0 || Nsig == 0) return GSPX_OK;
  int64_t max_ld = std::min<int64_t>(ops_max_ld(g, sizeof(T), 5), 256);
  if (max_ld < 1) return set_err(GSPX_ERR_INVALID, "graph too large: one signal column exceeds 2 GiB");
  const int* perm = g->has_perm ? g->perm.as<int>() : nullptr;
  const int* iperm = g->has_perm ? g->iperm.as<int>() : nullptr;
  const int nbN = std::max(1, (int)((N + 255) / 256));
  // A = tau L + diag(M) on the internal layout
  DevMem mint, aval, scal, partial;
  CHK(mint.alloc((size_t)N * sizeof(T)));
  CHK(aval.alloc(((size_t)g->nnz_int + 64) * sizeof(T)));
  HIPCHK(hipMemsetAsync(aval.p, 0, ((size_t)g->nnz_int + 64) * sizeof(T), st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  CHK(permute_panel<T>(g, mask, 1, mint.as<T>(), 1, perm));
  hipLaunchKernelGGL((k_affine_values<T>), dim3(nbN), dim3(256), 0, st, g->rptr.as<int>(),
                     g->rcol.as<int>(), g->rval.as<T>(), (int)N, (T)tau, mint.as<T>(), aval.as<T>());
  const int nred = (int)std::min<int64_t>(4096, std::max<int64_t>(1, N / 64));
  for (int64_t c0 = 0; c0 < Nsig; c0 += max_ld) {
    const unsigned ld = (unsigned)std::min<int64_t>(max_ld, Nsig - c0);
    int ldp = 1;
    while (ldp < (int)ld) ldp <<= 1;
    const size_t U = (size_t)N * ld;
    CHK(ctx->ws_t.ensure(5 * U * sizeof(T) + 256));
    T* X = ctx->ws_t.as<T>();
    T* R = X + U;
    T* P = R + U;
    T* Q = P + U;
    T* B = Q + U;
    CHK(scal.ensure((size_t)ld * (7 * sizeof(double) + 2 * sizeof(int)) + 64));
    CHK(partial.ensure((size_t)nred * ld * sizeof(double)));
    CgScalars s;
    double* d = scal.as<double>();
    s.rho_prev = d; s.rho_cur = d + ld; s.pq = d + 2 * ld; s.atol = d + 3 * ld; s.alpha = d + 4 * ld;
    s.beta = d + 5 * ld;
    double* rr = d + 6 * ld;
    s.active = (int*)(d + 7 * ld);
    s.iters = s.active + ld;
    s.any_active = s.iters + ld;
    const unsigned nbU = (unsigned)std::min<size_t>((U + 255) / 256, 65536);
    const int nbl = (int)((ld + 63) / 64);
    auto coldot = [&](const T* a_, const T* b_, double* out) {
      hipLaunchKernelGGL((k_coldot_partial<T>), dim3(nred), dim3(256), 0, st, a_, b_, (int)N, (int)ld,
                         ldp, partial.as<double>());
      hipLaunchKernelGGL(k_colsum, dim3(ld), dim3(64), 0, st, partial.as<double>(), nred, (int)ld, out);
    };
    // b = M y (learning.py:325-326 zeroes the unmeasured entries), r = b, x = 0
    CHK(permute_panel<T>(g, y + c0, (unsigned)Nsig, B, ld, perm));
    hipLaunchKernelGGL((k_rowscale<T>), dim3(nbU), dim3(256), 0, st, mint.as<T>(), B, R, U, (int)ld);
    HIPCHK(hipMemsetAsync(X, 0, U * sizeof(T), st));
    HIPCHK(hipMemsetAsync(P, 0, U * sizeof(T), st));
    coldot(R, R, rr);
    hipLaunchKernelGGL(k_cg_init, dim3(nbl), dim3(64), 0, st, s, rr, (int)ld, rtol, atol);
    for (int64_t it = 0; it < maxiter; ++it) {
      // (rr = ||r||^2 per column: from the coldot above for the first iteration, afterwards accumulated by the
      // update kernel of the previous one - k_cg_xr_dot - in the very same order)
      HIPCHK(hipMemsetAsync(s.any_active, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_cg_pre, dim3(nbl), dim3(64), 0, st, s, rr, (int)ld, it == 0 ? 1 : 0);
      // every fourth iteration the host looks whether any column is still active (a device-to-host copy and a
      // stream synchronisation: ~30 us of a 0.4 ms iteration); in between the device carries on - converged columns
      // are frozen by their `active` flag, so up to three iterations at the end do nothing and change nothing
      if (it < 2 || (it & 3) == 0 || it + 1 == maxiter) {
        int any = 0;
        HIPCHK(hipMemcpyAsync(&any, s.any_active, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!any) break;
      }
      hipLaunchKernelGGL((k_cg_p<T>), dim3(nbU), dim3(256), 0, st, R, P, U, (int)ld, s);
      CHK(spmm_internal<T>(g, aval.as<T>(), T(1), T(0), P, Q, ld, nullptr, 0));
      coldot(P, Q, s.pq);
      hipLaunchKernelGGL(k_cg_post, dim3(nbl), dim3(64), 0, st, s, s.pq, (int)ld);
      hipLaunchKernelGGL((k_cg_xr_dot<T>), dim3(nred), dim3(256), 0, st, X, R, P, Q, (int)N, (int)ld, ldp, s,
                         partial.as<double>());
      hipLaunchKernelGGL(k_colsum, dim3(ld), dim3(64), 0, st, partial.as<double>(), nred, (int)ld, rr);
    }
    CHK(permute_panel<T>(g, X, ld, x + c0, (unsigned)Nsig, iperm));
    if (iters) {
      std::vector<int> hi(ld);
      HIPCHK(hipMemcpyAsync(hi.data(), s.iters, ld * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      for (unsigned c = 0; c < ld; ++c) iters[c0 + c] = hi[c];
    }
  }
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, ctx->ev[0], ctx->ev[1]));
  if (ms) *ms = f;
  return GSPX_OK;
}

extern "C" int gspx_tikhonov_cg_dev(gspx_graph* g, double tau, const void* mask_dev, int64_t Nsig,
                                    const void* y_dev, void* x_dev, double rtol, double atol,
                                    int64_t maxiter, int32_t* iterations, double* kernel_ms) {
  if (g) replay_reset(g->ctx);
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  if (!(tau > 0) || !std::isfinite(tau)) return set_err(GSPX_ERR_INVALID, "tau must be positive and finite");
  if (Nsig < 0 || maxiter < 0 || !(rtol >= 0) || !(atol >= 0))
    return set_err(GSPX_ERR_INVALID, "tikhonov_cg: bad argument");
  if (Nsig > 0 && g->N > 0 && (!mask_dev || !y_dev || !x_dev))
    return set_err(GSPX_ERR_INVALID, "null pointer");
  HIPCHK(hipSetDevice(g->ctx->device));
  return g->dtype == GSPX_F32
             ? tikhonov_t<float>(g, tau, (const float*)mask_dev, Nsig, (const float*)y_dev,
                                 (float*)x_dev, rtol, atol, maxiter, iterations, kernel_ms)
             : tikhonov_t<double>(g, tau, (const double*)mask_dev, Nsig, (const double*)y_dev,
                                  (double*)x_dev, rtol, atol, maxiter, iterations, kernel_ms);
}

// ---- differential operator (undirected graphs, no self loops) ------------------------------------------
template <typename T> static int ensure_edges(gspx_graph* g) {
  if (g->edges_built) return GSPX_OK;
  if (!g->from_w)
    return set_err(GSPX_ERR_INVALID, "differential operator needs a graph created from W (degrees)");
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int N = (int)g->N;
  CHK(g->e_off.alloc(((size_t)N + 1) * sizeof(int)));
  CHK(g->e_toff.alloc(((size_t)N + 1) * sizeof(int)));
  if (N == 0) {
    g->n_edges = 0;
    g->edges_built = true;
    return GSPX_OK;
  }
  DevMem up, low;
  CHK(up.alloc(((size_t)N + 1) * sizeof(int)));
  CHK(low.alloc(((size_t)N + 1) * sizeof(int)));
  HIPCHK(hipMemsetAsync(up.p, 0, ((size_t)N + 1) * sizeof(int), st));
  HIPCHK(hipMemsetAsync(low.p, 0, ((size_t)N + 1) * sizeof(int), st));
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(k_edge_count, dim3(nb), dim3(256), 0, st, g->lptr.as<int>(), g->lcol.as<int>(), N,
                     up.as<int>(), low.as<int>());
  CHK(scan_exclusive(ctx, up.as<int>(), g->e_off.as<int>(), N + 1));
  CHK(scan_exclusive(ctx, low.as<int>(), g->e_toff.as<int>(), N + 1));
  int E = 0, El = 0;
  HIPCHK(hipMemcpyAsync(&E, g->e_off.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&El, g->e_toff.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (E != El) return set_err(GSPX_ERR_INVALID, "differential operator: the Laplacian pattern is not symmetric");
  const size_t e = (size_t)std::max(E, 1);
  CHK(g->e_src.alloc(e * sizeof(int)));
  CHK(g->e_dst.alloc(e * sizeof(int)));
  CHK(g->e_tedge.alloc(e * sizeof(int)));
  CHK(g->e_cs.alloc(e * sizeof(T)));
  CHK(g->e_ct.alloc(e * sizeof(T)));
  CHK(g->e_w.alloc(e * sizeof(T)));
  hipLaunchKernelGGL((k_edge_fill<T>), dim3(nb), dim3(256), 0, st, g->lptr.as<int>(), g->lcol.as<int>(),
                     g->lval.as<T>(), g->dw.as<T>(), N, g->lap_type, g->e_off.as<int>(),
                     g->e_toff.as<int>(), g->e_src.as<int>(), g->e_dst.as<int>(), g->e_cs.as<T>(),
                     g->e_ct.as<T>(), g->e_w.as<T>(), g->e_tedge.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  g->n_edges = E;
  g->edges_built = true;
  return GSPX_OK;
}

// ---- differential operator from a caller's edge list (directed graphs, self-loops) --------------------
// Graph.get_edge_list of a directed graph is every stored entry of W (graph.py:1019-1029) and its D carries a
// factor 1 / sqrt(2) (difference.py:160-161); with self-loops the diagonal entries are edges too, whose two D
// values cancel (difference.py:166 drops the stored zeros).  The device graph holds the Laplacian of the
// SYMMETRISED W, so these edges cannot be read back from it: the host layer hands over (source, target, weight)
// in get_edge_list order - sources non-decreasing - and the D values are formed here from the graph's degrees.
template <typename T>
static int set_edge_list_t(gspx_graph* g, int64_t E, const int32_t* src, const int32_t* dst, const double* w,
                           int directed) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int N = (int)g->N;
  std::vector<int> eoff((size_t)N + 1, 0), toff((size_t)N + 1, 0), tedge((size_t)std::max<int64_t>(E, 1));
  for (int64_t k = 0; k < E; ++k) {
    if (src[k] < 0 || src[k] >= N || dst[k] < 0 || dst[k] >= N)
      return set_err(GSPX_ERR_INVALID, "gspx_graph_set_edge_list: edge %lld has a vertex outside [0, N)", (long long)k);
    if (k > 0 && src[k] < src[k - 1])
      return set_err(GSPX_ERR_INVALID, "gspx_graph_set_edge_list: sources must not decrease (get_edge_list order)");
    if (!(w[k] >= 0.0)) return set_err(GSPX_ERR_INVALID, "gspx_graph_set_edge_list: weights must be >= 0");
    ++eoff[(size_t)src[k] + 1];
    ++toff[(size_t)dst[k] + 1];
  }
  for (int i = 0; i < N; ++i) {
    eoff[(size_t)i + 1] += eoff[i];
    toff[(size_t)i + 1] += toff[i];
  }
  {
    std::vector<int> cur(toff.begin(), toff.end() - 1);
    for (int64_t k = 0; k < E; ++k) tedge[(size_t)cur[dst[k]]++] = (int)k;  // edges ending in a vertex, in edge order
  }
  std::vector<T> wt((size_t)std::max<int64_t>(E, 1));
  for (int64_t k = 0; k < E; ++k) wt[(size_t)k] = (T)w[k];
  const size_t e = (size_t)std::max<int64_t>(E, 1);
  // the buffers are replaced one by one (alloc releases the old one first): until the last copy is queued the graph
  // has NO edge list, so that a failure half-way leaves grad / div to rebuild the default one instead of running on
  // freed or mismatched buffers
  g->edges_built = false;
  g->n_edges = 0;
  CHK(g->e_off.alloc(((size_t)N + 1) * sizeof(int)));
  CHK(g->e_toff.alloc(((size_t)N + 1) * sizeof(int)));
  CHK(g->e_src.alloc(e * sizeof(int)));
  CHK(g->e_dst.alloc(e * sizeof(int)));
  CHK(g->e_tedge.alloc(e * sizeof(int)));
  CHK(g->e_cs.alloc(e * sizeof(T)));
  CHK(g->e_ct.alloc(e * sizeof(T)));
  CHK(g->e_w.alloc(e * sizeof(T)));
  HIPCHK(hipMemcpyAsync(g->e_off.p, eoff.data(), ((size_t)N + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(g->e_toff.p, toff.data(), ((size_t)N + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  if (E > 0) {
    HIPCHK(hipMemcpyAsync(g->e_src.p, src, (size_t)E * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(g->e_dst.p, dst, (size_t)E * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(g->e_tedge.p, tedge.data(), (size_t)E * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(g->e_w.p, wt.data(), (size_t)E * sizeof(T), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_edge_values<T>), dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, g->e_src.as<int>(),
                       g->e_dst.as<int>(), g->e_w.as<T>(), g->dw.as<T>(), (int)E, g->lap_type, directed,
                       g->e_cs.as<T>(), g->e_ct.as<T>());
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(st));  // the host vectors above are read by the copies until here
  g->n_edges = E;
  g->edges_built = true;
  return GSPX_OK;
}

extern "C" int gspx_graph_set_edge_list(gspx_graph* g, int64_t n_edges, const int32_t* sources, const int32_t* targets,
                                        const double* weights, int directed) {
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  if (n_edges < 0 || n_edges >= ((int64_t)1 << 31) || (n_edges > 0 && (!sources || !targets || !weights)))
    return set_err(GSPX_ERR_INVALID, "gspx_graph_set_edge_list: bad argument");
  if (!g->from_w)
    return set_err(GSPX_ERR_INVALID, "differential operator needs a graph created from W (degrees)");
  HIPCHK(hipSetDevice(g->ctx->device));
  return g->dtype == GSPX_F32 ? set_edge_list_t<float>(g, n_edges, sources, targets, weights, directed)
                              : set_edge_list_t<double>(g, n_edges, sources, targets, weights, directed);
}

static int edges_for(gspx_graph* g) {
  if (!g) return set_err(GSPX_ERR_INVALID, "null graph");
  HIPCHK(hipSetDevice(g->ctx->device));
  return g->dtype == GSPX_F32 ? ensure_edges<float>(g) : ensure_edges<double>(g);
}

extern "C" int gspx_graph_n_edges(gspx_graph* g, int64_t* n_edges) {
  if (!n_edges) return set_err(GSPX_ERR_INVALID, "null argument");
  CHK(edges_for(g));
  *n_edges = g->n_edges;
  return GSPX_OK;
}

extern "C" int gspx_graph_download_edges(gspx_graph* g, int32_t* sources, int32_t* targets,
                                         void* weights, void* d_source, void* d_target) {
  CHK(edges_for(g));
  const size_t E = (size_t)g->n_edges, e = elt_size(g->dtype);
  if (E == 0) return GSPX_OK;
  if (sources) HIPCHK(hipMemcpy(sources, g->e_src.p, E * sizeof(int), hipMemcpyDeviceToHost));
  if (targets) HIPCHK(hipMemcpy(targets, g->e_dst.p, E * sizeof(int), hipMemcpyDeviceToHost));
  if (weights) HIPCHK(hipMemcpy(weights, g->e_w.p, E * e, hipMemcpyDeviceToHost));
  if (d_source) HIPCHK(hipMemcpy(d_source, g->e_cs.p, E * e, hipMemcpyDeviceToHost));
  if (d_target) HIPCHK(hipMemcpy(d_target, g->e_ct.p, E * e, hipMemcpyDeviceToHost));
  return GSPX_OK;
}

template <typename T>
static int grad_div_t(gspx_graph* g, bool is_div, int64_t Nsig, const T* in, T* out, double* ms) {
  gspx_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  if (ms) *ms = 0;
  const size_t rows = (size_t)(is_div ? g->N : g->n_edges);
  if (rows == 0 || Nsig == 0) return GSPX_OK;
  int cw = 1;  // lanes across the signals of one row: a power of two up to 64
  while (cw < 64 && cw < Nsig) cw <<= 1;
  const unsigned nb = (unsigned)std::min<size_t>((rows + (256 / cw) - 1) / (256 / cw), 1 << 20);
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  constexpr int TV = 16 / (int)sizeof(T);
  if ((Nsig % TV) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && g->N >= 8 &&
      ctx->opt.edge_vertex_walk) {
    // vertex walk in the internal order (see k_grad_v): lane groups of 16-byte lanes, one per vertex
    int gs = 1;
    while (gs < 64 && gs < Nsig / TV) gs <<= 1;
    const int per_xcd = (int)((g->N + 7) / 8), gpb = 256 / gs;
    const unsigned nbx = (unsigned)std::min<int64_t>(((int64_t)per_xcd + gpb - 1) / gpb, 8192);
    const int* perm = g->has_perm ? g->perm.as<int>() : nullptr;
    if (is_div)
      hipLaunchKernelGGL((k_div_v<T, TV>), dim3(nbx * 8), dim3(256), 0, st, perm, g->e_off.as<int>(),
                         g->e_toff.as<int>(), g->e_tedge.as<int>(), g->e_cs.as<T>(), g->e_ct.as<T>(), in, out,
                         (int)g->N, (int)Nsig, gs, per_xcd);
    else
      hipLaunchKernelGGL((k_grad_v<T, TV>), dim3(nbx * 8), dim3(256), 0, st, perm, g->e_off.as<int>(),
                         g->e_dst.as<int>(), g->e_cs.as<T>(), g->e_ct.as<T>(), in, out, (int)g->N, (int)Nsig, gs,
                         per_xcd);
  } else if (is_div)
    hipLaunchKernelGGL((k_div<T>), dim3(nb), dim3(256), 0, st, g->e_off.as<int>(), g->e_toff.as<int>(),
                       g->e_tedge.as<int>(), g->e_cs.as<T>(), g->e_ct.as<T>(), in, out, (int)g->N,
                       (int)Nsig, cw);
  else
    hipLaunchKernelGGL((k_grad<T>), dim3(nb), dim3(256), 0, st, g->e_src.as<int>(), g->e_dst.as<int>(),
                       g->e_cs.as<T>(), g->e_ct.as<T>(), in, out, (size_t)g->n_edges, (int)Nsig, cw);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, ctx->ev[0], ctx->ev[1]));
  if (ms) *ms = f;
  return GSPX_OK;
}

static int grad_div(gspx_graph* g, bool is_div, int64_t Nsig, const void* in, void* out, double* ms) {
  CHK(edges_for(g));
  if (Nsig < 0 || Nsig > (1 << 20)) return set_err(GSPX_ERR_INVALID, "bad number of signals");
  if (Nsig > 0 && (!in || !out) && g->N > 0 && g->n_edges > 0)
    return set_err(GSPX_ERR_INVALID, "null signal pointer");
  return g->dtype == GSPX_F32 ? grad_div_t<float>(g, is_div, Nsig, (const float*)in, (float*)out, ms)
                              : grad_div_t<double>(g, is_div, Nsig, (const double*)in, (double*)out, ms);
}

extern "C" int gspx_grad_dev(gspx_graph* g, int64_t Nsig, const void* x_dev, void* y_dev,
                             double* kernel_ms) {
  return grad_div(g, false, Nsig, x_dev, y_dev, kernel_ms);
}
extern "C" int gspx_div_dev(gspx_graph* g, int64_t Nsig, const void* y_dev, void* z_dev,
                            double* kernel_ms) {
  return grad_div(g, true, Nsig, y_dev, z_dev, kernel_ms);
}
