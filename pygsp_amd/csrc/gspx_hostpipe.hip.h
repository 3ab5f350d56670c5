// gspx_hostpipe.hip.h - the host-pointer entry point (gspx_cheby_filter) as a five-stage pipeline over
// signal-column batches.  Included by gspx.hip.
//
// What a plugin-mode caller of pygsp/filters/filter.py:146-328 hands over is pageable numpy memory: round 2
// copied the whole input (pageable hipMemcpyAsync: the runtime stages it on one thread), ran the kernels,
// copied the whole output back - 18 + 11 + 18 ms for the 1M x 64 fp64 headline call.  The columns of a panel
// are independent (the recurrence never mixes signals), so the call is cut into column batches and the
// stages overlap on three streams and two small groups of host threads:
//
//   pack(b+2)    host threads gather columns [c0, c0+w) of x (row pitch Nsig) into a PINNED compact panel
//   H2D(b+1)     one contiguous DMA of that panel                                   (stream_in)
//   kernels(b)   the ordinary device path on the compact panel (ld = w)              (the context's stream)
//   D2H(b-1)     one contiguous DMA of the compact result into pinned memory        (stream_out)
//   unpack(b-2)  host threads scatter it into columns [c0, c0+w) of y
//
// Three buffers on the input side (slot = b % 3: packing and shipping batch b overlaps the kernels of batches b-2
// AND b-1 - with two, pack + H2D of a batch, 5 ms for 16 fp64 columns, had to fit behind ONE batch's kernels, 3 ms,
// and the kernels starved), two on the output side (slot = b & 1).  All hand-offs are host-side (a mutex, a condition variable and four
// counters), plus ONE device-side dependency: the kernels of batch b wait for the H2D event of batch b on
// their stream, so the compute thread never blocks on a copy.  The staging buffers are pinned once per
// context and grow only.  Per column the arithmetic is the device path's own, so the result equals the
// unpipelined call bit for bit (tests/test_gpu_2_kernels.py::test_host_pipeline_*).
#pragma once

#include <sys/mman.h>

#include <condition_variable>
#include <thread>

// (PinMem and HostPipe - the pinned staging buffers, streams and events a context keeps for this - are defined
// in gspx.hip next to the context itself)

// rows [r0, r1) of a strided column block <-> a compact panel; `rb` bytes per row piece
static inline void copy_rows(unsigned char* dst, size_t dpitch, const unsigned char* src, size_t spitch, size_t rb,
                             int64_t r0, int64_t r1) {
  dst += (size_t)r0 * dpitch;
  src += (size_t)r0 * spitch;
  switch (rb) {  // the common piece sizes get an inlined fixed-size copy
#define GSPX_CR(B)                                                        \
  case B:                                                                 \
    for (int64_t r = r0; r < r1; ++r, dst += dpitch, src += spitch) memcpy(dst, src, B); \
    return;
    GSPX_CR(32)
    GSPX_CR(64)
    GSPX_CR(128)
    GSPX_CR(256)
#undef GSPX_CR
    default:
      for (int64_t r = r0; r < r1; ++r, dst += dpitch, src += spitch) memcpy(dst, src, rb);
  }
}

#define GSPX_HOSTPIPE_UNAVAILABLE (-1) /* internal: the pipeline could not get its staging buffers */

// `widths`: the signal columns of each batch, in order (they add up to Nsig)
template <typename T>
static int filter_host_pipelined(gspx_graph* g, double lmax, int Nf, int M, const double* coeffs, int64_t Nsig,
                                 const T* x, T* y, int mode, const std::vector<int64_t>& widths, int nthreads,
                                 double* kernel_ms) {
  gspx_ctx* ctx = g->ctx;
  HostPipe& hp = *ctx->pipe;
  if (hp.init() != GSPX_OK) {
    (void)hipGetLastError();
    return GSPX_HOSTPIPE_UNAVAILABLE;
  }
  const int64_t N = g->N;
  const bool analysis = mode == GSPX_ANALYSIS;
  const int in_planes = analysis ? 1 : Nf, out_planes = analysis ? Nf : 1;
  const int nb = (int)widths.size();
  std::vector<int64_t> col0((size_t)nb, 0);
  for (int b = 1; b < nb; ++b) col0[(size_t)b] = col0[(size_t)b - 1] + widths[(size_t)b - 1];
  const int64_t w = *std::max_element(widths.begin(), widths.end());
  {  // an in-place call (y overlapping x): the one-shot form reads all of x before it writes y and is safe; the
     // pipeline writes - and pre-faults - columns of y while later columns of x are still to be read
    const uintptr_t xa = (uintptr_t)x, xb = xa + (size_t)in_planes * N * Nsig * sizeof(T);
    const uintptr_t ya = (uintptr_t)y, yb = ya + (size_t)out_planes * N * Nsig * sizeof(T);
    if (xa < yb && ya < xb) return GSPX_HOSTPIPE_UNAVAILABLE;
  }
  {  // the result array is usually fresh from the allocator (mmap): ask for huge pages before the unpacking
     // threads fault it in (131k page faults of 4 KB for a 512 MB result cost them several milliseconds)
    const size_t bytes = (size_t)out_planes * N * Nsig * sizeof(T);
    const uintptr_t lo = ((uintptr_t)y + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
    const uintptr_t hi = ((uintptr_t)y + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
  }
  const size_t in_cap = (size_t)in_planes * N * w * sizeof(T), out_cap = (size_t)out_planes * N * w * sizeof(T);
  constexpr int NIN = HostPipe::NIN;
  for (int s = 0; s < NIN; ++s) {
    // (no staging memory - pinned or device - is not an error of the call: the caller falls back to the one-shot form)
    if (hp.pin_in[s].ensure(in_cap) != GSPX_OK || hp.dx[s].ensure(in_cap) != GSPX_OK ||
        (s < 2 && (hp.pin_out[s].ensure(out_cap) != GSPX_OK || hp.dy[s].ensure(out_cap) != GSPX_OK))) {
      (void)hipGetLastError();
      return GSPX_HOSTPIPE_UNAVAILABLE;
    }
  }
  const int P = std::max(1, nthreads), Q = std::max(1, nthreads);
  const auto wall0 = std::chrono::steady_clock::now();

  std::mutex mu;
  std::condition_variable cv;
  int issued = 0, computed = 0, shipped = 0, unpacked = 0;  // batches past each stage
  std::vector<int> pack_arrived((size_t)nb, 0), unpack_arrived((size_t)nb, 0);
  int err = GSPX_OK;
  std::string err_msg;
  auto fail = [&](int rc) {  // called with g_err set on the calling thread
    std::lock_guard<std::mutex> lock(mu);
    if (err == GSPX_OK) {
      err = rc;
      err_msg = g_err;
    }
    cv.notify_all();
  };
  auto wait_for = [&](auto pred) {  // false: the pipeline failed elsewhere
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [&] { return err != GSPX_OK || pred(); });
    return err == GSPX_OK;
  };
  auto hipfail = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return false;
    fail(set_err(GSPX_ERR_HIP, "%s failed in the host pipeline: %s", what, hipGetErrorString(e)));
    return true;
  };
  std::vector<double> pack_ms((size_t)P, 0.0), unpack_ms((size_t)Q, 0.0);
  std::vector<double> line((size_t)nb * 6, 0.0);  // (each entry is written by the one thread that completes the stage)
  auto stamp = [&](int b, int what) {
    line[(size_t)b * 6 + what] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  };
  double h2d_ms = 0, d2h_ms = 0;
  auto width_of = [&](int b) { return widths[(size_t)b]; };

  auto pack_worker = [&](int p) {
    const int64_t r0 = N * p / P, r1 = N * (p + 1) / P;
    for (int b = 0; b < nb; ++b) {
      const int s = b % NIN;
      if (!wait_for([&] { return computed >= b - (NIN - 1); })) return;  // slot s: batch b - NIN consumed
      const auto t0 = std::chrono::steady_clock::now();
      const int64_t wl = width_of(b);
      for (int pl = 0; pl < in_planes; ++pl)
        copy_rows((unsigned char*)hp.pin_in[s].p + (size_t)pl * N * wl * sizeof(T), (size_t)wl * sizeof(T),
                  (const unsigned char*)(x + (size_t)pl * N * Nsig + (size_t)col0[(size_t)b]), (size_t)Nsig * sizeof(T),
                  (size_t)wl * sizeof(T), r0, r1);
      pack_ms[(size_t)p] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      bool last;
      {
        std::lock_guard<std::mutex> lock(mu);
        last = ++pack_arrived[(size_t)b] == P;
      }
      if (last) {  // the panel is complete: one contiguous DMA
        stamp(b, 0);
        if (hipfail(hipSetDevice(ctx->device), "hipSetDevice")) return;
        if (b >= NIN) {  // the slot's previous DMA (batch b - NIN) is complete: its kernels have run
          float t = 0;
          if (hipEventElapsedTime(&t, hp.t_in[s][0], hp.t_in[s][1]) == hipSuccess) h2d_ms += t;
        }
        if (hipfail(hipEventRecord(hp.t_in[s][0], hp.stream_in), "hipEventRecord")) return;
        if (hipfail(hipMemcpyAsync(hp.dx[s].p, hp.pin_in[s].p, (size_t)in_planes * N * wl * sizeof(T),
                                   hipMemcpyHostToDevice, hp.stream_in), "hipMemcpyAsync (H2D)")) return;
        if (hipfail(hipEventRecord(hp.t_in[s][1], hp.stream_in), "hipEventRecord")) return;
        if (hipfail(hipEventRecord(hp.h2d_ev[s], hp.stream_in), "hipEventRecord")) return;
        stamp(b, 1);
        std::lock_guard<std::mutex> lock(mu);
        issued = b + 1;
        cv.notify_all();
      }
    }
  };
  auto shipper = [&]() {
    if (hipfail(hipSetDevice(ctx->device), "hipSetDevice")) return;
    for (int b = 0; b < nb; ++b) {
      const int s = b & 1;
      if (!wait_for([&] { return computed >= b + 1 && unpacked >= b - 1; })) return;
      const int64_t wl = width_of(b);
      if (hipfail(hipEventRecord(hp.t_out[0], hp.stream_out), "hipEventRecord")) return;
      if (hipfail(hipMemcpyAsync(hp.pin_out[s].p, hp.dy[s].p, (size_t)out_planes * N * wl * sizeof(T),
                                 hipMemcpyDeviceToHost, hp.stream_out), "hipMemcpyAsync (D2H)")) return;
      if (hipfail(hipEventRecord(hp.t_out[1], hp.stream_out), "hipEventRecord")) return;
      if (hipfail(hipStreamSynchronize(hp.stream_out), "hipStreamSynchronize")) return;
      float t = 0;
      if (hipEventElapsedTime(&t, hp.t_out[0], hp.t_out[1]) == hipSuccess) d2h_ms += t;
      stamp(b, 4);
      std::lock_guard<std::mutex> lock(mu);
      shipped = b + 1;
      cv.notify_all();
    }
  };
  auto unpack_worker = [&](int q) {
    const int64_t r0 = N * q / Q, r1 = N * (q + 1) / Q;
    {  // While the first batch is still being packed, shipped and computed this thread has nothing to do: it faults
       // its rows of the result in now (y is usually fresh from the allocator; it is output-only, so touching it is
       // harmless), which takes the page faults and the kernel's page zeroing off the critical path.
      for (int pl = 0; pl < out_planes; ++pl) {
        volatile unsigned char* base = (volatile unsigned char*)(y + (size_t)pl * N * Nsig);
        const size_t lo = (size_t)r0 * Nsig * sizeof(T), hi = (size_t)r1 * Nsig * sizeof(T);
        for (size_t off = lo; off < hi; off += 4096) base[off] = 0;
        if (hi > lo) base[hi - 1] = 0;
      }
    }
    for (int b = 0; b < nb; ++b) {
      const int s = b & 1;
      if (!wait_for([&] { return shipped >= b + 1; })) return;
      const auto t0 = std::chrono::steady_clock::now();
      const int64_t wl = width_of(b);
      for (int pl = 0; pl < out_planes; ++pl)
        copy_rows((unsigned char*)(y + (size_t)pl * N * Nsig + (size_t)col0[(size_t)b]), (size_t)Nsig * sizeof(T),
                  (const unsigned char*)hp.pin_out[s].p + (size_t)pl * N * wl * sizeof(T), (size_t)wl * sizeof(T),
                  (size_t)wl * sizeof(T), r0, r1);
      unpack_ms[(size_t)q] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::lock_guard<std::mutex> lock(mu);
      if (++unpack_arrived[(size_t)b] == Q) {
        stamp(b, 5);
        unpacked = b + 1;
        cv.notify_all();
      }
    }
  };

  std::vector<std::thread> threads;
  threads.reserve((size_t)P + Q + 1);
  try {
    for (int p = 0; p < P; ++p) threads.emplace_back(pack_worker, p);
    threads.emplace_back(shipper);
    for (int q = 0; q < Q; ++q) threads.emplace_back(unpack_worker, q);
  } catch (...) {
    fail(set_err(GSPX_ERR_HIP, "could not start the host pipeline threads"));
  }

  // this thread: the kernels, batch by batch
  double k_ms = 0, tm[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nb; ++b) {
    const int s = b % NIN, so = b & 1;
    if (!wait_for([&] { return issued >= b + 1 && shipped >= b - 1; })) break;
    if (hipfail(hipStreamWaitEvent(ctx->stream, hp.h2d_ev[s], 0), "hipStreamWaitEvent")) break;
    stamp(b, 2);
    const BatchProgram* bp = ctx->batch_program;  // (gspx_poly_program: a polynomial program on every batch)
    const int rc = bp ? program_dev_t<T>(g, lmax, bp->S, bp->sc, bp->be, bp->ga, bp->old_is_x, width_of(b),
                                         (const T*)hp.dx[s].p, (T*)hp.dy[so].p)
                      : filter_dev_t<T>(g, lmax, Nf, M, coeffs, width_of(b), (const T*)hp.dx[s].p, (T*)hp.dy[so].p, mode);
    if (rc != GSPX_OK) {
      fail(rc);
      break;
    }
    stamp(b, 3);
    k_ms += ctx->timing[0];
    for (int i = 0; i < 5; ++i) tm[i] += ctx->timing[i];
    std::lock_guard<std::mutex> lock(mu);
    computed = b + 1;
    cv.notify_all();
  }
  for (auto& t : threads) t.join();
  (void)hipSetDevice(ctx->device);
  if (err != GSPX_OK) {
    (void)hipStreamSynchronize(hp.stream_in);
    (void)hipStreamSynchronize(hp.stream_out);
    (void)hipStreamSynchronize(ctx->stream);
    g_err = err_msg;
    return err;
  }
  for (int b = std::max(0, nb - NIN); b < nb; ++b) {  // the DMAs whose slot was not reused
    float t = 0;
    if (hipEventElapsedTime(&t, hp.t_in[b % NIN][0], hp.t_in[b % NIN][1]) == hipSuccess) h2d_ms += t;
  }
  (void)hipGetLastError();
  for (int i = 0; i < 5; ++i) ctx->timing[i] = tm[i];
  hp.timing[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  hp.timing[1] = *std::max_element(pack_ms.begin(), pack_ms.end());
  hp.timing[2] = h2d_ms;
  hp.timing[3] = k_ms;
  hp.timing[4] = d2h_ms;
  hp.timing[5] = *std::max_element(unpack_ms.begin(), unpack_ms.end());
  hp.timing[6] = nb;
  hp.timing[7] = (double)w;
  hp.timing[8] = P;
  hp.timeline = line;
  if (kernel_ms) *kernel_ms = k_ms;
  return GSPX_OK;
}

// batches and threads of a pipelined call; no batches: run the one-shot path
static void host_pipeline_shape(const Options& opt, size_t elt, int64_t N, int64_t Nsig, int planes_total,
                                std::vector<int64_t>* widths, int* threads_out) {
  widths->clear();
  if (!opt.host_pipeline) return;
  const size_t total = (size_t)N * (size_t)Nsig * elt * (size_t)planes_total;
  if (opt.host_pipeline == 1 && total < ((size_t)48 << 20)) return;  // too small to pay for the threads
  // The call's critical path is: fill (pack + H2D of the first batch), the kernels of all batches, drain (D2H +
  // unpack of the last batch).  Batches of 128-byte rows (16 fp64 / 32 fp32 signals) run the kernels within ~20 %
  // of their full-width rate (the matrix is streamed once per batch; narrower panels cost much more per column)
  // while a batch's transfer still hides behind the previous batch's kernels; the first and the last batch are
  // half as wide, which halves fill and drain.  (Measured on the 1M x 64 fp64 call, profiles/r03_hostpipe_*:
  // uniform 8 / 16 / 32 columns 28.5 / 25.0 / 31.2 ms; narrow edges followed by 3x wider batches starve the
  // kernels - the second batch's transfer outlasts the first batch's kernels - 27.5 ms.)
  const int64_t unit = (int64_t)(64 / elt);  // auto sizes are multiples of 64-byte rows: the kernels' efficient widths
  int64_t w = opt.host_batch > 0 ? opt.host_batch : (int64_t)(128 / elt);
  if (opt.host_batch <= 0 && Nsig < 4 * w) w = std::max<int64_t>(unit, (Nsig / 4) / unit * unit);
  w = std::min<int64_t>(w, Nsig);
  int64_t edge = opt.host_edge > 0 ? std::min<int64_t>(opt.host_edge, w) : (opt.host_batch > 0 ? w : std::max(unit, w / 2 / unit * unit));
  if (Nsig <= w || (opt.host_pipeline == 1 && Nsig < 2 * w)) {
    if (opt.host_pipeline != 2 || Nsig < 2) return;
    w = edge = (Nsig + 1) / 2;  // "always": two halves
  }
  if (2 * edge >= Nsig) edge = w;  // too few columns for separate edge batches
  if (edge < w) {
    widths->push_back(edge);
    for (int64_t c = edge; c < Nsig - edge; c += w) widths->push_back(std::min<int64_t>(w, Nsig - edge - c));
    widths->push_back(edge);
  } else {
    for (int64_t c = 0; c < Nsig; c += w) widths->push_back(std::min<int64_t>(w, Nsig - c));
  }
  // a ragged tail of less than 32-byte rows would run the sub-wave kernel, which sums a row's entries in another
  // order than the tile / panel kernels: merged into its neighbour, so that the automatic schedule returns the
  // bytes of the one-shot call
  if (opt.host_batch <= 0 && widths->size() >= 2 && (size_t)widths->back() * elt < 32) {
    const int64_t tail = widths->back();
    widths->pop_back();
    widths->back() += tail;
  }
  int t = (int)opt.host_threads;
  if (t <= 0) {
    const unsigned hc = std::thread::hardware_concurrency();
    t = (int)std::min<unsigned>(16, std::max<unsigned>(1, hc / 4));
  }
  *threads_out = std::min(t, 64);
}
