// gspx_comm.hip.h - the path's one collective inside the library: the gather of the ranks' outputs to a
// root over RCCL (xGMI), SURVEY.md 8(b)/(e).  Included by gspx.hip.
//
// The recurrence itself never communicates (independent graphs / signal columns per GPU); only the final
// outputs travel.  Two forms:
//   * one process per GPU (bench.py --gpus N under torch.distributed.run): gspx_comm_create() from a
//     128-byte RCCL unique id that the launcher hands from rank 0 to the others, then gspx_comm_gather();
//   * one process driving several GPUs from threads: gspx_gather() builds its communicators with
//     ncclCommInitAll over the devices of the parts.
// Every block, the root's own included, travels as a grouped ncclSend / ncclRecv pair, so each peer's
// block lands over its own xGMI link and the n transfers run concurrently (a ring collective would be
// bound by one link).  RCCL is loaded on first use (dlopen): libgspx has no link-time dependency on it
// and single-GPU users never pay for it.  The reference has no multi-device path; nothing is replaced.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace gspx_rccl {

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;              // optional (reporting only)
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  std::string why;  // why loading failed
};

static Api& instance() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (!a.lib) {
      const char* e = dlerror();
      a.why = e ? e : "librccl.so.1 not found";
      return;
    }
    bool ok = true;
    auto sym = [&](const char* n) {
      void* p = dlsym(a.lib, n);
      if (!p) {
        ok = false;
        a.why = std::string("missing RCCL symbol ") + n;
      }
      return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommInitAll = (decltype(a.CommInitAll))sym("ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
      dlclose(a.lib);
      a.lib = nullptr;
      return;
    }
    a.GetVersion = (decltype(a.GetVersion))dlsym(a.lib, "ncclGetVersion");
    a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.lib, "ncclCommUserRank");
  });
  return a;
}
static Api* api() {
  Api& a = instance();
  return a.lib ? &a : nullptr;
}
static std::string unavailable() { return "RCCL is not available: " + instance().why; }

}  // namespace gspx_rccl

#define RCCLCHK(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return set_err(GSPX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, R->GetErrorString(r_), __FILE__, \
                     __LINE__);                                                                    \
  } while (0)

struct gspx_comm {
  gspx_ctx* ctx = nullptr;  // null once the context was destroyed: the handle is then only good for gspx_comm_destroy
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
};

// gspx_ctx_destroy: every communicator still alive on the context is torn down with it (its stream is about to
// go) and its handle is left as an empty shell, so a later gspx_comm_gather / gspx_comm_destroy on it is an error
// message / a plain delete instead of a use-after-free.
static void comm_invalidate_all(gspx_ctx* ctx) {
  std::vector<gspx_comm*> live;
  {
    std::lock_guard<std::mutex> lock(ctx->comms_mu);
    live.swap(ctx->comms);
  }
  gspx_rccl::Api* R = live.empty() ? nullptr : gspx_rccl::api();
  for (gspx_comm* h : live) {
    if (R && h->comm) {
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      (void)R->CommDestroy(h->comm);
    }
    h->comm = nullptr;
    h->ctx = nullptr;
  }
}

extern "C" int gspx_comm_available(void) { return gspx_rccl::api() ? 1 : 0; }

extern "C" int gspx_comm_unique_id(unsigned char id[GSPX_COMM_ID_BYTES]) {
  static_assert(GSPX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return set_err(GSPX_ERR_INVALID, "gspx_comm_unique_id: null argument");
  gspx_rccl::Api* R = gspx_rccl::api();
  if (!R) return set_err(GSPX_ERR_HIP, "%s", gspx_rccl::unavailable().c_str());
  ncclUniqueId u;
  RCCLCHK(R->GetUniqueId(&u));
  memcpy(id, u.internal, GSPX_COMM_ID_BYTES);
  return GSPX_OK;
}

extern "C" int gspx_comm_create(gspx_ctx* ctx, int nranks, int rank, const unsigned char id[GSPX_COMM_ID_BYTES],
                                gspx_comm** out) {
  if (!ctx || !out || !id || nranks < 1 || rank < 0 || rank >= nranks)
    return set_err(GSPX_ERR_INVALID, "gspx_comm_create: bad argument");
  *out = nullptr;
  gspx_rccl::Api* R = gspx_rccl::api();
  if (!R) return set_err(GSPX_ERR_HIP, "%s", gspx_rccl::unavailable().c_str());
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(u.internal, id, GSPX_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  RCCLCHK(R->CommInitRank(&c, nranks, u, rank));
  gspx_comm* h = new gspx_comm();
  h->ctx = ctx;
  h->rank = rank;
  h->nranks = nranks;
  h->comm = c;
  {
    std::lock_guard<std::mutex> lock(ctx->comms_mu);
    ctx->comms.push_back(h);
  }
  *out = h;
  return GSPX_OK;
}

extern "C" int gspx_comm_destroy(gspx_comm* h) {
  if (!h) return GSPX_OK;
  if (h->ctx) {  // (a null context: gspx_ctx_destroy already took the communicator down)
    {
      std::lock_guard<std::mutex> lock(h->ctx->comms_mu);
      auto& v = h->ctx->comms;
      v.erase(std::remove(v.begin(), v.end(), h), v.end());
    }
    gspx_rccl::Api* R = gspx_rccl::api();
    if (R && h->comm) {
      (void)hipSetDevice(h->ctx->device);
      (void)hipStreamSynchronize(h->ctx->stream);
      (void)R->CommDestroy(h->comm);
    }
  }
  delete h;
  return GSPX_OK;
}

// Every rank calls this with the same `bytes` table (bytes[r] = size of rank r's block).  The root receives
// the blocks back to back, in rank order, into root_out (which must hold their sum); other ranks pass
// root_out = NULL.  Queued on the context's stream behind the filtering that produced `part`; returns
// when this rank's transfers have completed.
extern "C" int gspx_comm_gather(gspx_comm* h, const void* part_dev, const int64_t* bytes, int root,
                                void* root_out_dev, double* ms) {
  if (!h || !bytes || root < 0 || root >= h->nranks)
    return set_err(GSPX_ERR_INVALID, "gspx_comm_gather: bad argument");
  if (!h->ctx || !h->comm)
    return set_err(GSPX_ERR_INVALID, "gspx_comm_gather: the communicator's context was destroyed");
  for (int r = 0; r < h->nranks; ++r)
    if (bytes[r] < 0) return set_err(GSPX_ERR_INVALID, "gspx_comm_gather: negative block size");
  if (bytes[h->rank] > 0 && !part_dev) return set_err(GSPX_ERR_INVALID, "gspx_comm_gather: null part");
  if (h->rank == root && !root_out_dev) return set_err(GSPX_ERR_INVALID, "gspx_comm_gather: the root needs an output");
  gspx_rccl::Api* R = gspx_rccl::api();
  if (!R) return set_err(GSPX_ERR_HIP, "%s", gspx_rccl::unavailable().c_str());
  gspx_ctx* ctx = h->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  RCCLCHK(R->GroupStart());
  ncclResult_t bad = ncclSuccess;  // a group that was opened is always closed
  auto note = [&](ncclResult_t r) {
    if (bad == ncclSuccess) bad = r;
  };
  if (bytes[h->rank] > 0) note(R->Send(part_dev, (size_t)bytes[h->rank], ncclChar, root, h->comm, st));
  if (h->rank == root) {
    size_t off = 0;
    for (int r = 0; r < h->nranks; ++r) {
      if (bytes[r] > 0) note(R->Recv((char*)root_out_dev + off, (size_t)bytes[r], ncclChar, r, h->comm, st));
      off += (size_t)bytes[r];
    }
  }
  note(R->GroupEnd());
  RCCLCHK(bad);
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipStreamSynchronize(st));
  if (ms) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
    *ms = t;
  }
  return GSPX_OK;
}

// ---- single-process form: several contexts (GPUs) driven by one process -----------------------------
namespace gspx_rccl {
// communicators of a device set, made once by ncclCommInitAll (rank i = i-th device of the sorted set)
// They live until the process exits: destroying RCCL communicators from a static destructor would race the
// HIP runtime's own teardown.  One gather at a time per set (`busy`): the communicators and the grouped
// send / recv of a set are not re-entrant.
struct DeviceSet {
  std::vector<int> devs;
  std::vector<ncclComm_t> comms;
  std::mutex busy;
};
static std::mutex g_sets_mu;
static std::atomic<int> g_last_set_ranks{0};  // ranks of the device set the last RCCL gspx_gather ran on (0: none yet)
static std::vector<std::unique_ptr<DeviceSet>> g_sets;

static DeviceSet* device_set(Api* R, const std::vector<int>& devs) {
  std::lock_guard<std::mutex> lock(g_sets_mu);
  for (auto& s : g_sets)
    if (s->devs == devs) return s.get();
  std::unique_ptr<DeviceSet> s(new DeviceSet());
  s->devs = devs;
  s->comms.assign(devs.size(), nullptr);
  if (R->CommInitAll(s->comms.data(), (int)devs.size(), devs.data()) != ncclSuccess) {
    for (ncclComm_t c : s->comms)  // whatever a partial initialisation left behind
      if (c) (void)R->CommDestroy(c);
    return nullptr;
  }
  g_sets.push_back(std::move(s));
  return g_sets.back().get();
}
}  // namespace gspx_rccl

// RCCL form of gspx_gather: GSPX_OK when it ran, any other code (with the error set) when it could not -
// the caller then falls back to peer copies.  `force`: also route same-device blocks through RCCL
// (self send / recv); without it a gather whose parts all live on the root's device needs no RCCL at all.
static int gather_rccl(int n, gspx_buf** parts, gspx_buf* root_out, bool force) {
  gspx_rccl::Api* R = gspx_rccl::api();
  if (!R) return set_err(GSPX_ERR_HIP, "%s", gspx_rccl::unavailable().c_str());
  gspx_ctx* root = root_out->ctx;
  std::vector<int> devs{root->device};
  for (int i = 0; i < n; ++i) devs.push_back(parts[i]->ctx->device);
  std::sort(devs.begin(), devs.end());
  devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
  if (devs.size() == 1 && !force) return set_err(GSPX_ERR_INVALID, "gather_rccl: one device, nothing to send");
  gspx_rccl::DeviceSet* set = gspx_rccl::device_set(R, devs);
  if (!set) return set_err(GSPX_ERR_HIP, "ncclCommInitAll failed");
  std::lock_guard<std::mutex> one_at_a_time(set->busy);
  auto rank_of = [&](int dev) { return (int)(std::lower_bound(devs.begin(), devs.end(), dev) - devs.begin()); };
  // one stream per device: the first context seen on it; everything queued on the parts' streams so far
  // must be complete before a foreign stream reads them
  std::vector<gspx_ctx*> lead(devs.size(), nullptr);
  lead[(size_t)rank_of(root->device)] = root;
  for (int i = 0; i < n; ++i) {
    gspx_ctx* c = parts[i]->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!lead[(size_t)rank_of(c->device)]) lead[(size_t)rank_of(c->device)] = c;
  }
  HIPCHK(hipSetDevice(root->device));
  HIPCHK(hipStreamSynchronize(root->stream));
  const int rr = rank_of(root->device);
  // same-device blocks (unless forced through RCCL): plain device copies, outside the group
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const size_t nb = (size_t)parts[i]->bytes;
    if (nb && rank_of(parts[i]->ctx->device) == rr && !force) {
      HIPCHK(hipSetDevice(root->device));
      HIPCHK(hipMemcpyAsync((char*)root_out->mem.p + off, parts[i]->mem.p, nb, hipMemcpyDeviceToDevice, root->stream));
    }
    off += parts[i]->bytes;
  }
  RCCLCHK(R->GroupStart());
  ncclResult_t bad = ncclSuccess;  // a group that was opened is always closed
  auto note = [&](ncclResult_t r) {
    if (bad == ncclSuccess) bad = r;
  };
  off = 0;
  for (int i = 0; i < n; ++i) {
    const size_t nb = (size_t)parts[i]->bytes;
    const int sr = rank_of(parts[i]->ctx->device);
    if (nb && (sr != rr || force)) {
      (void)hipSetDevice(devs[(size_t)sr]);
      note(R->Send(parts[i]->mem.p, nb, ncclChar, rr, set->comms[(size_t)sr], lead[(size_t)sr]->stream));
      (void)hipSetDevice(root->device);
      note(R->Recv((char*)root_out->mem.p + off, nb, ncclChar, sr, set->comms[(size_t)rr], root->stream));
    }
    off += parts[i]->bytes;
  }
  note(R->GroupEnd());
  RCCLCHK(bad);
  {  // what the communicator itself says it spans (gspx_comm_info): not what this function was asked for
    int cnt = (int)devs.size();
    if (R->CommCount && R->CommCount(set->comms[(size_t)rr], &cnt) != ncclSuccess) cnt = -1;
    gspx_rccl::g_last_set_ranks.store(cnt);
  }
  for (size_t d = 0; d < devs.size(); ++d) {
    HIPCHK(hipSetDevice(devs[d]));
    HIPCHK(hipStreamSynchronize(lead[d]->stream));
  }
  HIPCHK(hipSetDevice(root->device));
  return GSPX_OK;
}

// Reporting: what RCCL itself says.  out[0] = RCCL version code (ncclGetVersion; 0 when RCCL cannot be loaded or
// does not export it), out[1] = ranks of the communicator as the communicator reports them (ncclCommCount; for
// h == NULL: of the device set the last RCCL gspx_gather of this process ran on, 0 when there was none),
// out[2] = this handle's rank in it (ncclCommUserRank; -1 for h == NULL).
extern "C" int gspx_comm_info(gspx_comm* h, int64_t out[3]) {
  if (!out) return set_err(GSPX_ERR_INVALID, "gspx_comm_info: null output");
  out[0] = out[1] = 0;
  out[2] = -1;
  gspx_rccl::Api* R = gspx_rccl::api();
  if (!R) return GSPX_OK;
  int v = 0;
  if (R->GetVersion && R->GetVersion(&v) == ncclSuccess) out[0] = v;
  if (!h) {
    out[1] = gspx_rccl::g_last_set_ranks.load();
    return GSPX_OK;
  }
  if (!h->ctx || !h->comm) return set_err(GSPX_ERR_INVALID, "gspx_comm_info: the communicator's context was destroyed");
  int cnt = h->nranks, rk = h->rank;
  if (R->CommCount) RCCLCHK(R->CommCount(h->comm, &cnt));
  if (R->CommUserRank) RCCLCHK(R->CommUserRank(h->comm, &rk));
  out[1] = cnt;
  out[2] = rk;
  return GSPX_OK;
}
