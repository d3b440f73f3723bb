// Multi-GPU exchange step of the windowed BA inside the library.
//
// Every accumulator of the window is a plain sum over points, and the reference sums per-thread partial copies before it
// stitches (src/OptimizationBackend/AccumulatedTopHessian.cpp:299-308, AccumulatedSCHessian.cpp:136-185).  With the points of a
// window sharded over ranks the same sum runs across GPUs: ONE ncclAllReduce(sum, float32) over xGMI of the packed accumulator
// block [topA | topL | accD | accE | accEB | Hcc | bc | nres] of every window of the batch (contiguous: one collective however many
// windows), enqueued on the context's stream between sdso_ba_batch_accumulate and sdso_ba_batch_solve; every rank then stitches and
// solves the same system.  A C++ FullSystem needs nothing but these entry points — no Python, no torch.
//
// RCCL is resolved with dlopen at sdso_comm_init (librccl.so.1; a process that already carries an RCCL, e.g. through
// torch.distributed, gets that copy): libsdso_hip.so itself has no link-time dependency on it and single-GPU users never load it.
#include "sdso_internal.h"
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include <map>
#include <memory>
#include <vector>
#include <algorithm>
#include <mutex>
#include <string>
#include <cstdlib>

namespace sdso {
void* ba_batch_accum_block(sdso_ctx* ctx, size_t* nfloats);            // ba.hip
void* ba_window_accum_block(sdso_ctx* ctx, int win, size_t* nfloats);  // ba.hip
bool ba_batch_scatter_wanted(sdso_ctx* ctx);                           // ba.hip: the batch's exchange is the reduce-scatter by window (agreed on by all ranks at optimize_begin)
void ba_batch_scatter_done(sdso_ctx* ctx);                             // ba.hip: ... and it has been enqueued: only this rank's windows hold summed accumulators

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// SDSO_RCCL_LIB overrides the library name (a deployment with its own RCCL build; tests force the "not loadable" path with it)
static RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    const char* forced = getenv("SDSO_RCCL_LIB");
    std::vector<const char*> names;
    if (forced && *forced) names = {forced};
    else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string last;
    for (const char* name : names) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
      const char* e = dlerror();            // (one call: dlerror clears the message it returns)
      last = e ? e : "dlopen failed";
    }
    if (!api.lib) err = std::string("RCCL not found: ") + last;
    else {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
      api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
      api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.lib, "ncclReduceScatter");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather || !api.ReduceScatter) { err = "RCCL symbols missing"; api.lib = nullptr; }
    }
  });
  if (!api.lib) { if (why) *why = err; return nullptr; }
  return &api;
}

// one communicator may serve several contexts of a process (e.g. the two stream groups of bench.py): collectives on one
// communicator are issued in the same order on every rank, whatever stream they run on
struct Comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  // host transport (sdso_comm_init_host): the collectives are staged through host memory and handed to the caller's functions
  sdso_host_allreduce_fn h_allreduce = nullptr;
  sdso_host_allgather_fn h_allgather = nullptr;
  void* h_user = nullptr;
  std::vector<float> h_send, h_recv;
  std::mutex h_mutex;       // the staging vectors are shared by every ctx attached to this communicator
  bool host() const { return h_allreduce != nullptr; }
  ~Comm() { if (comm) { RcclApi* a = rccl_api(nullptr); if (a) a->CommDestroy(comm); } }
};
static std::map<sdso_ctx*, std::shared_ptr<Comm>> g_comms;
void release_comm(sdso_ctx* ctx) {
  std::shared_ptr<Comm> dying;     // destroyed (ncclCommDestroy may block) after the registry lock is released
  {
    std::lock_guard<std::mutex> g(registry_mutex());
    auto it = g_comms.find(ctx);
    if (it != g_comms.end()) { dying = std::move(it->second); g_comms.erase(it); }
  }
}
static std::shared_ptr<Comm> comm_of(sdso_ctx* ctx) {
  std::lock_guard<std::mutex> g(registry_mutex());
  auto it = g_comms.find(ctx);
  return it == g_comms.end() ? nullptr : it->second;
}
}  // namespace sdso

using namespace sdso;

namespace sdso {
// used by the resident GN loop (ba.hip): ranks of ctx's communicator (1 without one), the all-gather of the per-rank energy / break-test
// records, and the max over ranks of a host int (collective, synchronises the ctx stream)
int comm_nranks(sdso_ctx* ctx) { auto c = comm_of(ctx); return c ? c->nranks : 1; }
int comm_rank(sdso_ctx* ctx) { auto c = comm_of(ctx); return c ? c->rank : 0; }
bool comm_present(sdso_ctx* ctx) { return comm_of(ctx) != nullptr; }
int comm_allgather_floats(sdso_ctx* ctx, const float* send, float* recv, size_t nfloats) {
  auto c = comm_of(ctx);
  if (!c) return sdso::fail(ctx, SDSO_ERR_STATE, "no communicator");
  if (c->host()) {
    std::lock_guard<std::mutex> hg(c->h_mutex);
    c->h_send.resize(nfloats); c->h_recv.resize(nfloats * c->nranks);
    SDSO_HIP(ctx, hipMemcpyAsync(c->h_send.data(), send, sizeof(float) * nfloats, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (c->h_allgather(c->h_user, c->h_send.data(), c->h_recv.data(), nfloats)) return sdso::fail(ctx, SDSO_ERR_STATE, "the host transport's all-gather failed");
    SDSO_HIP(ctx, hipMemcpyAsync(recv, c->h_recv.data(), sizeof(float) * nfloats * c->nranks, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SDSO_OK;
  }
  RcclApi* a = rccl_api(nullptr);
  const ncclResult_t r = a->AllGather(send, recv, nfloats, ncclFloat32, c->comm, ctx->stream);
  if (r != ncclSuccess) return sdso::fail(ctx, SDSO_ERR_HIP, std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "RCCL error"));
  return SDSO_OK;
}
int comm_max_int(sdso_ctx* ctx, int* value) {
  auto c = comm_of(ctx);
  if (!c) return SDSO_OK;
  if (c->host()) {
    const float mine = (float)*value;                       // capacities are far below 2^24: exact
    std::vector<float> all(c->nranks);
    if (c->h_allgather(c->h_user, &mine, all.data(), 1)) return sdso::fail(ctx, SDSO_ERR_STATE, "the host transport's all-gather failed");
    for (float v : all) *value = std::max(*value, (int)v);
    return SDSO_OK;
  }
  RcclApi* a = rccl_api(nullptr);
  int* d = nullptr;
  if (hipMalloc(&d, sizeof(int)) != hipSuccess) return sdso::fail(ctx, SDSO_ERR_HIP, "hipMalloc");
  hipMemcpyAsync(d, value, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
  const ncclResult_t r = a->AllReduce(d, d, 1, ncclInt32, ncclMax, c->comm, ctx->stream);
  hipMemcpyAsync(value, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  hipFree(d);
  if (r != ncclSuccess || e != hipSuccess) return sdso::fail(ctx, SDSO_ERR_HIP, "all-reduce(max) of the pack capacity failed");
  return SDSO_OK;
}
}  // namespace sdso

#define SDSO_NCCL(ctx, api, expr)                                                                                              \
  do {                                                                                                                         \
    ncclResult_t _r = (expr);                                                                                                  \
    if (_r != ncclSuccess) return sdso::fail(ctx, SDSO_ERR_HIP, std::string(#expr) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(_r) : "RCCL error")); \
  } while (0)

extern "C" int sdso_comm_unique_id(void* id128) {
  if (!id128) return SDSO_ERR_ARG;
  RcclApi* a = rccl_api(nullptr);
  if (!a) return SDSO_ERR_STATE;
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return SDSO_ERR_HIP;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return SDSO_OK;
}

extern "C" int sdso_comm_init(sdso_ctx* ctx, int nranks, int rank, const void* id128) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks && id128, "bad communicator arguments");
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  std::string why;
  RcclApi* a = rccl_api(&why);
  if (!a) return sdso::fail(ctx, SDSO_ERR_STATE, why);
  release_comm(ctx);
  auto c = std::make_shared<Comm>();
  c->nranks = nranks; c->rank = rank; c->device = ctx->device;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  SDSO_NCCL(ctx, a, a->CommInitRank(&c->comm, nranks, id, rank));
  std::lock_guard<std::mutex> g(registry_mutex());
  g_comms[ctx] = c;
  return SDSO_OK;
}

// Bring-your-own transport (MPI, sockets, a test harness): the same collectives, staged through host memory and carried out by the
// caller's functions.  Every library path that talks to the communicator (sdso_ba_allreduce*, the resident loop's all-gather) works on
// top of it; it synchronises the ctx stream around each collective, so it is the slow path by construction.
extern "C" int sdso_comm_init_host(sdso_ctx* ctx, int nranks, int rank, sdso_host_allreduce_fn allreduce, sdso_host_allgather_fn allgather, void* user) {
  if (!ctx) return SDSO_ERR_STATE;
  SDSO_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks && allreduce && allgather, "bad communicator arguments");
  release_comm(ctx);
  auto c = std::make_shared<Comm>();
  c->nranks = nranks; c->rank = rank; c->device = ctx->device;
  c->h_allreduce = allreduce; c->h_allgather = allgather; c->h_user = user;
  std::lock_guard<std::mutex> g(registry_mutex());
  g_comms[ctx] = c;
  return SDSO_OK;
}

extern "C" int sdso_comm_attach(sdso_ctx* ctx, sdso_ctx* owner) {
  if (!ctx || !owner) return SDSO_ERR_STATE;
  auto c = comm_of(owner);
  SDSO_REQUIRE(ctx, c, "the owner context has no communicator");
  SDSO_REQUIRE(ctx, c->device == ctx->device, "contexts that share a communicator must sit on the same device");
  std::lock_guard<std::mutex> g(registry_mutex());
  g_comms[ctx] = c;
  return SDSO_OK;
}

extern "C" int sdso_comm_info(sdso_ctx* ctx, int* nranks, int* rank) {
  if (!ctx) return SDSO_ERR_STATE;
  auto c = comm_of(ctx);
  if (nranks) *nranks = c ? c->nranks : 0;
  if (rank) *rank = c ? c->rank : -1;
  return SDSO_OK;
}

extern "C" int sdso_comm_destroy(sdso_ctx* ctx) {
  if (!ctx) return SDSO_ERR_STATE;
  hipStreamSynchronize(ctx->stream);
  release_comm(ctx);
  return SDSO_OK;
}

static int allreduce_block(sdso_ctx* ctx, void* ptr, size_t nfloats) {
  auto c = comm_of(ctx);
  SDSO_REQUIRE(ctx, c, "no communicator: call sdso_comm_init (or sdso_comm_attach) first");
  SDSO_REQUIRE(ctx, ptr && nfloats > 0, "nothing to reduce");
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  if (c->host()) {
    std::lock_guard<std::mutex> hg(c->h_mutex);
    c->h_send.resize(nfloats);
    SDSO_HIP(ctx, hipMemcpyAsync(c->h_send.data(), ptr, sizeof(float) * nfloats, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (c->h_allreduce(c->h_user, c->h_send.data(), nfloats)) return sdso::fail(ctx, SDSO_ERR_STATE, "the host transport's all-reduce failed");
    SDSO_HIP(ctx, hipMemcpyAsync(ptr, c->h_send.data(), sizeof(float) * nfloats, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SDSO_OK;
  }
  RcclApi* a = rccl_api(nullptr);
  SDSO_NCCL(ctx, a, a->AllReduce(ptr, ptr, nfloats, ncclFloat32, ncclSum, c->comm, ctx->stream));
  return SDSO_OK;
}

// The other shape of the same exchange (sdso_ba_batch_exchange_mode(ctx, 1)): the windows of the batch lie one after the other in the block,
// so a reduce-scatter hands rank r the SUMMED accumulators of windows [r * nwin / N, (r + 1) * nwin / N) and nothing of the others — half
// the bytes of the all-reduce on every xGMI link; the solve of a window then runs on one rank only and x comes back by all-gather
// (ba.hip: opt_solve_step).  In place: rank r's slice of the block is both its send and its receive segment.
static int reduce_scatter_block(sdso_ctx* ctx, void* ptr, size_t nfloats) {
  auto c = comm_of(ctx);
  SDSO_REQUIRE(ctx, c, "no communicator: call sdso_comm_init (or sdso_comm_attach) first");
  SDSO_REQUIRE(ctx, ptr && nfloats > 0 && nfloats % c->nranks == 0, "the block does not divide over the ranks");
  SDSO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t per = nfloats / c->nranks;
  if (c->host()) {   // the host transport sums every slice; only this rank's comes back, so the rest of the block is left as RCCL leaves it
    std::lock_guard<std::mutex> hg(c->h_mutex);
    c->h_send.resize(nfloats);
    SDSO_HIP(ctx, hipMemcpyAsync(c->h_send.data(), ptr, sizeof(float) * nfloats, hipMemcpyDeviceToHost, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (c->h_allreduce(c->h_user, c->h_send.data(), nfloats)) return sdso::fail(ctx, SDSO_ERR_STATE, "the host transport's all-reduce failed");
    SDSO_HIP(ctx, hipMemcpyAsync((float*)ptr + per * c->rank, c->h_send.data() + per * c->rank, sizeof(float) * per, hipMemcpyHostToDevice, ctx->stream));
    SDSO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SDSO_OK;
  }
  RcclApi* a = rccl_api(nullptr);
  SDSO_NCCL(ctx, a, a->ReduceScatter(ptr, (float*)ptr + per * c->rank, per, ncclFloat32, ncclSum, c->comm, ctx->stream));
  return SDSO_OK;
}

extern "C" int sdso_ba_allreduce(sdso_ctx* ctx) {
  if (!ctx) return SDSO_ERR_STATE;
  size_t n = 0;
  void* p = ba_batch_accum_block(ctx, &n);
  SDSO_REQUIRE(ctx, p, "no batch: sdso_ba_batch_create first");
  ProfScope ps(ctx, "sdso_ba_allreduce", 2);   // (level-2 profiling: HIP events around the exchange on the ctx stream — bench.py's extra.exchange_ms_per_step)
  if (ba_batch_scatter_wanted(ctx)) {
    const int rc = reduce_scatter_block(ctx, p, n);
    if (rc == SDSO_OK) ba_batch_scatter_done(ctx);      // a failed exchange leaves the batch as it was (sdso_ba_batch_solve still refuses nothing)
    return rc;
  }
  return allreduce_block(ctx, p, n);
}

extern "C" int sdso_ba_allreduce_window(sdso_ctx* ctx, int win) {
  if (!ctx) return SDSO_ERR_STATE;
  size_t n = 0;
  void* p = ba_window_accum_block(ctx, win, &n);
  SDSO_REQUIRE(ctx, p, "unknown window");
  return allreduce_block(ctx, p, n);
}
