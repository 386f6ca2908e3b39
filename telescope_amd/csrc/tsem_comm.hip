// libtelescope_em.so, collectives: RCCL resolved at RUN time (never linked), the in-process transport (several engines on one
// device, tests), and the communicator ABI (tsem_comm_*).  SURVEY 8(e): one sum all-reduce of K+2 doubles per EM iteration.
#include "tsem_internal.h"

#include <dlfcn.h>
#include <link.h>

#include <condition_variable>
#include <mutex>

extern "C" {

// ---------------------------------------------------------------------------
// collectives: RCCL resolved at RUN time, or the in-process transport
// ---------------------------------------------------------------------------
// RCCL is not linked.  A torch process already carries a librccl (torch/lib/librccl.so, loaded with torch); linking a
// second one by DT_NEEDED made the copy that serves this library's calls depend on load order.
// Now ONE copy is chosen deliberately: the librccl that is already mapped into the process if there is one (so the
// library and torch.distributed share a single RCCL — one set of IPC handles, one topology detection), otherwise
// librccl.so.1 from the loader's search path / /opt/rocm/lib.  A box without RCCL can still load and run the library
// on one GPU; tsem_comm_library_info reports which copy and version is in use (it goes into the bench line).
struct NcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool ok = false;
  int version = 0;
  std::string path, others, err;
};
static int nccl_phdr_cb(struct dl_phdr_info* info, size_t, void* data) {
  auto* v = static_cast<std::vector<std::string>*>(data);
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) v->push_back(info->dlpi_name);
  return 0;
}
static NcclApi* nccl_api() {
  static NcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> loaded;
    dl_iterate_phdr(nccl_phdr_cb, &loaded);
    void* hnd = nullptr;
    if (!loaded.empty()) {
      hnd = dlopen(loaded[0].c_str(), RTLD_NOW | RTLD_NOLOAD);
      if (hnd) api.path = loaded[0];
      for (size_t i = 1; i < loaded.size(); ++i) api.others += (api.others.empty() ? "" : ", ") + loaded[i];
    }
    const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 3 && !hnd; ++i) {
      hnd = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
      if (hnd) {
        api.path = cands[i];
        Dl_info di;
        void* sym = dlsym(hnd, "ncclAllReduce");
        if (sym && dladdr(sym, &di) && di.dli_fname) api.path = di.dli_fname;
      }
    }
    if (!hnd) { api.err = std::string("librccl is not available: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    auto need = [&](const char* name) -> void* {
      void* p = dlsym(hnd, name);
      if (!p && api.err.empty()) api.err = std::string("librccl (") + api.path + ") does not export " + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(need("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(need("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(need("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(need("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(need("ncclGetErrorString"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(need("ncclGetVersion"));
    if (!api.err.empty()) return;
    if (api.GetVersion(&api.version) != ncclSuccess) api.version = 0;
    // the id / enum layout this file was compiled against (rccl.h of the ROCm image) is the 2.x ABI
    if (api.version && api.version / 10000 != NCCL_MAJOR) {
      api.err = "librccl (" + api.path + ") has major version " + std::to_string(api.version / 10000) + ", this library was built for " +
                std::to_string(NCCL_MAJOR);
      return;
    }
    api.ok = true;
  });
  return &api;
}

// ---- in-process transport -------------------------------------------------------------------------------------
// Several handles on ONE device, each driven by its own host thread of one process, run the protocol of a row-sharded
// job: same tsem_em_chunk, same reduce-buffer layout, same device-side stop flag and error slot — only the all-reduce
// itself is different.  Rank r copies its vector into its slot, records an event; after a HOST rendezvous (every rank has
// recorded) each rank makes its stream wait for the peers' events and sums the slots in rank order, so all ranks get the
// same bits.  Two slot generations alternate; a slot is rewritten only after the peers' sums of two collectives ago
// have completed (their `done` events).  For tests of the N > 1 path on a one-GPU box, and for hosts that time-slice
// one GPU between several engines; a multi-GPU job uses RCCL.
constexpr int TS_LOCAL_MAXW = 8;
struct tsem_local_group {
  int world = 1, device = 0, refs = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  double timeout_s = 120.0;
  void* slot[TS_LOCAL_MAXW][2] = {};
  size_t slot_bytes[TS_LOCAL_MAXW][2] = {};
  hipEvent_t ready[TS_LOCAL_MAXW][2] = {}, done[TS_LOCAL_MAXW][2] = {};
  bool done_rec[TS_LOCAL_MAXW][2] = {};
  bool taken[TS_LOCAL_MAXW] = {};
};
// host rendezvous of the group's ranks; false: a peer did not arrive in time (or the group broke earlier)
static bool local_rendezvous(tsem_local_group* g) {
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->broken) return false;
  const uint64_t my = g->gen;
  if (++g->arrived == g->world) { g->arrived = 0; ++g->gen; g->cv.notify_all(); return true; }
  const bool ok = g->cv.wait_for(lk, std::chrono::duration<double>(g->timeout_s), [&] { return g->gen != my || g->broken; });
  if (!ok || g->broken) { g->broken = true; g->cv.notify_all(); return false; }
  return true;
}
// dtype / op as in tsem_comm_allreduce_host: 0 f64 sum, 1 u64 sum, 2 f64 max, 3 i64 max
struct LocalSlots { const void* p[TS_LOCAL_MAXW]; };
__global__ __launch_bounds__(256) void k_local_reduce(void* __restrict__ out, LocalSlots S, int world, int64_t n, int dtype) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dtype == 0 || dtype == 2) {
    double v = static_cast<const double*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) { const double t = static_cast<const double*>(S.p[q])[i]; v = dtype == 0 ? v + t : fmax(v, t); }
    static_cast<double*>(out)[i] = v;
  } else if (dtype == 1) {
    unsigned long long v = static_cast<const unsigned long long*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) v += static_cast<const unsigned long long*>(S.p[q])[i];
    static_cast<unsigned long long*>(out)[i] = v;
  } else {
    long long v = static_cast<const long long*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) v = max(v, static_cast<const long long*>(S.p[q])[i]);
    static_cast<long long*>(out)[i] = v;
  }
}
static int local_allreduce(tsem_comm* c, void* buf, size_t count, int dtype, hipStream_t s, std::string& err) {
  tsem_local_group* g = c->local;
  const int r = c->rank, par = (int)(c->epoch & 1);
  c->epoch += 1;
  const size_t bytes = count * 8;
  auto fail = [&](const std::string& m) { err = m; std::lock_guard<std::mutex> lk(g->mu); g->broken = true; g->cv.notify_all(); return TSEM_ERR_HIP; };
#define LOC_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
  // my slot of this generation is free once the peers' sums of two collectives ago are done
  for (int q = 0; q < g->world; ++q)
    if (q != r && g->done_rec[q][par]) LOC_HIP(hipStreamWaitEvent(s, g->done[q][par], 0));
  if (g->slot_bytes[r][par] < bytes) {
    for (int q = 0; q < g->world; ++q)
      if (q != r && g->done_rec[q][par]) LOC_HIP(hipEventSynchronize(g->done[q][par]));
    if (g->slot[r][par]) LOC_HIP(hipFree(g->slot[r][par]));
    g->slot[r][par] = nullptr; g->slot_bytes[r][par] = 0;
    LOC_HIP(hipMalloc(&g->slot[r][par], std::max<size_t>(bytes, 4096)));
    g->slot_bytes[r][par] = std::max<size_t>(bytes, 4096);
  }
  if (bytes) LOC_HIP(hipMemcpyAsync(g->slot[r][par], buf, bytes, hipMemcpyDeviceToDevice, s));
  LOC_HIP(hipEventRecord(g->ready[r][par], s));
  if (!local_rendezvous(g)) { err = "in-process communicator: a peer rank did not reach the collective (time-out or an earlier failure)"; return TSEM_ERR_TIMEOUT; }
  LocalSlots S;
  for (int q = 0; q < g->world; ++q) {
    if (q != r) LOC_HIP(hipStreamWaitEvent(s, g->ready[q][par], 0));
    S.p[q] = g->slot[q][par];
  }
  if (count) k_local_reduce<<<(unsigned)((count + 255) / 256), 256, 0, s>>>(buf, S, g->world, (int64_t)count, dtype);
  LOC_HIP(hipGetLastError());
  LOC_HIP(hipEventRecord(g->done[r][par], s));
  g->done_rec[r][par] = true;
#undef LOC_HIP
  return TSEM_OK;
}

bool tsem_comm_on(const tsem_ctx* h) { return h->comm && h->comm->active(); }
// in-place all-reduce of `count` 8-byte words on device memory, on stream s, over whichever transport the communicator has
int tsem_comm_allreduce_dev(tsem_comm* c, void* buf, size_t count, int dtype, hipStream_t s, std::string& err) {
  if (!c || !c->active()) return TSEM_OK;
  if (c->local) return local_allreduce(c, buf, count, dtype, s, err);
  NcclApi* N = nccl_api();
  const ncclDataType_t dt = dtype == 1 ? ncclUint64 : (dtype == 3 ? ncclInt64 : ncclDouble);
  const ncclRedOp_t op = dtype >= 2 ? ncclMax : ncclSum;
  const ncclResult_t r = N->AllReduce(buf, buf, count, dt, op, c->nccl, s);
  if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + N->GetErrorString(r); return TSEM_ERR_HIP; }
  return TSEM_OK;
}

// the per-iteration exchange: ONE sum all-reduce of the per-locus column sums + the error flag
int tsem_comm_allreduce_red(tsem_ctx* h, int64_t offset, int64_t count) {
  if (!tsem_comm_on(h)) return TSEM_OK;
  return tsem_comm_allreduce_dev(h->comm, h->d_red + offset, (size_t)count, 0, h->stream, h->err);
}
// ---------------------------------------------------------------------------
// communicator (RCCL over xGMI; one per process / GPU)
// ---------------------------------------------------------------------------
static thread_local std::string g_comm_err;            // (per host thread: the in-process transport runs one rank per thread)
const char* tsem_comm_last_error(void) { return g_comm_err.c_str(); }

int tsem_comm_library_info(char* buf, int32_t cap) {
  if (!buf || cap <= 0) return TSEM_ERR_ARG;
  NcclApi* N = nccl_api();
  std::string t;
  if (N->ok) {
    t = "rccl " + std::to_string(N->version / 10000) + "." + std::to_string(N->version / 100 % 100) + "." + std::to_string(N->version % 100) +
        " (" + N->path + ")";
    if (!N->others.empty()) t += "; other copies mapped: " + N->others;
  } else {
    t = "rccl unavailable: " + N->err;
  }
  snprintf(buf, (size_t)cap, "%s", t.c_str());
  return N->ok ? TSEM_OK : TSEM_ERR_HIP;
}

int tsem_comm_unique_id(void* id128) {
  if (!id128) return TSEM_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == TSEM_COMM_ID_BYTES, "ncclUniqueId size");
  NcclApi* N = nccl_api();
  if (!N->ok) { g_comm_err = N->err; return TSEM_ERR_HIP; }
  ncclUniqueId id;
  ncclResult_t r = N->GetUniqueId(&id);
  if (r != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + N->GetErrorString(r); return TSEM_ERR_HIP; }
  memcpy(id128, &id, sizeof(id));
  return TSEM_OK;
}

int tsem_comm_create(tsem_comm** out, int device, const void* id128, int rank, int world) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return TSEM_ERR_ARG;
  *out = nullptr;
  NcclApi* N = nccl_api();
  if (!N->ok) { g_comm_err = N->err; return TSEM_ERR_HIP; }
  if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  tsem_comm* c = new tsem_comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclResult_t r = N->CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + N->GetErrorString(r); delete c; return TSEM_ERR_HIP; }
  *out = c;
  return TSEM_OK;
}

int tsem_comm_local_group(tsem_local_group** out, int device, int world) {
  if (!out || world < 1 || world > TS_LOCAL_MAXW) return TSEM_ERR_ARG;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  tsem_local_group* g = new tsem_local_group();
  g->world = world; g->device = device; g->refs = 1;       // the creator's handle
  for (int q = 0; q < world; ++q)
    for (int par = 0; par < 2; ++par)
      if (hipEventCreateWithFlags(&g->ready[q][par], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g->done[q][par], hipEventDisableTiming) != hipSuccess) {
        g_comm_err = "hipEventCreate failed"; delete g; return TSEM_ERR_HIP;
      }
  *out = g;
  return TSEM_OK;
}

// The group lives until its creator's handle AND every communicator made from it are gone (`refs`): a caller that destroys the group
// while a rank's communicator still exists — an error path that skipped a rank's close, a garbage-collected Python object — used to
// leave that communicator with a dangling pointer, and its destructor locked a destroyed mutex (std::system_error, found by the
// row-sharded soak).  A destroyed group is `broken` for the ranks that still hold it: their collectives fail instead of waiting.
static void local_group_release(tsem_local_group* g) {
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (--g->refs > 0) return;
  }
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < g->world; ++q)
    for (int par = 0; par < 2; ++par) {
      if (g->slot[q][par]) (void)hipFree(g->slot[q][par]);
      if (g->ready[q][par]) (void)hipEventDestroy(g->ready[q][par]);
      if (g->done[q][par]) (void)hipEventDestroy(g->done[q][par]);
    }
  delete g;
}

void tsem_comm_local_group_destroy(tsem_local_group* g) {
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;                                      // (ranks still inside a collective, or entering one later, fail at once)
    g->cv.notify_all();
  }
  local_group_release(g);
}

int tsem_comm_create_local(tsem_comm** out, tsem_local_group* g, int rank) {
  if (!out || !g || rank < 0 || rank >= g->world) return TSEM_ERR_ARG;
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->taken[rank]) { g_comm_err = "tsem_comm_create_local: this rank of the group is taken"; return TSEM_ERR_ARG; }
    g->taken[rank] = true;
    g->refs += 1;
  }
  tsem_comm* c = new tsem_comm();
  c->device = g->device; c->rank = rank; c->world = g->world; c->local = g;
  *out = c;
  return TSEM_OK;
}

void tsem_comm_destroy(tsem_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->nccl) (void)nccl_api()->CommDestroy(c->nccl);
  if (c->local) {
    { std::lock_guard<std::mutex> lk(c->local->mu); c->local->taken[c->rank] = false; }
    local_group_release(c->local);
  }
  delete c;
}

int tsem_comm_attach(tsem_ctx* h, tsem_comm* c) {
  if (!h) return TSEM_ERR_ARG;
  if (c && c->device != h->device) TSEM_FAIL(TSEM_ERR_ARG, "communicator and handle are on different devices");
  h->comm = c;
  return TSEM_OK;
}

int tsem_comm_allreduce(tsem_ctx* h, int64_t offset, int64_t count) {
  if (!h || !h->have_model || offset < 0 || count < 0 || offset + count > h->K + 2) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return tsem_comm_allreduce_red(h, offset, count);
}

int tsem_comm_allreduce_host(tsem_comm* c, void* data, int64_t count, int dtype) {
  if (!c || !c->active() || (!data && count) || count < 0 || dtype < 0 || dtype > 3) return TSEM_ERR_ARG;
  if (count == 0) return TSEM_OK;
  if (hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  const size_t bytes = (size_t)count * 8;
  if (c->stage_bytes < bytes) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr; c->stage_bytes = 0;
    if (hipMalloc(&c->d_stage, bytes) != hipSuccess) { g_comm_err = "hipMalloc failed (all-reduce staging)"; return TSEM_ERR_NOMEM; }
    c->stage_bytes = bytes;
  }
  // Host vectors travel on the null stream.  The same communicator also serves an engine's own stream (tsem_em_chunk);
  // RCCL wants one stream per communicator at a time, so everything the device still has queued is drained first
  // (these are set-up and report sums: a device synchronisation costs nothing here).
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(c->d_stage, data, bytes, hipMemcpyHostToDevice);
  int rc = TSEM_OK;
  if (e == hipSuccess) rc = tsem_comm_allreduce_dev(c, c->d_stage, (size_t)count, dtype, nullptr, g_comm_err);
  if (rc) return rc;
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e == hipSuccess) e = hipMemcpy(data, c->d_stage, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { g_comm_err = std::string("all-reduce staging: ") + hipGetErrorString(e); return TSEM_ERR_HIP; }
  return TSEM_OK;
}


}  // extern "C"
