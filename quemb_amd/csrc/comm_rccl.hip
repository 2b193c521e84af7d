// comm_rccl.hip -- the one exchange step of the sharded fragment sweep, on a persistent RCCL communicator (gfx950 / xGMI).
//
// Reference: none -- QuEmb has no communication backend; be_func_parallel returns every worker's (e_f, mo_coeff, rdm1, rdm2s, rdm1_tmp)
// through pathos pipes (molbe/be_parallel.py:484-517) and solve_error then reads Fobjs[j]._rdm1 (molbe/solver.py:763).  SURVEY.md
// section 5 / 8(e): one process per GPU, fragments statically partitioned, ONE sum-all-reduce of a few hundred doubles per sweep ->
// latency bound; ncclAllReduce(ncclDouble, ncclSum) on a communicator created once.
//
// librccl is opened lazily (dlopen) by the first dev_comm_* call: a single-GPU user of libqemb_hip.so needs no RCCL on the machine, and a
// process that has already loaded an RCCL (the one bundled with PyTorch) shares that copy instead of starting a second one.
// All calls are made by the host thread that owns the DEFAULT execution context (the sweep joins its worker threads before the exchange);
// the collective runs on that context's stream, behind every kernel the rank has queued there.
//
// A lost peer is an ERROR, not a hang (round 4): the rendezvous (ncclCommInitRank blocks until every rank has joined) runs in a helper
// thread that the caller waits for at most QEMB_COMM_INIT_TIMEOUT_S seconds (default 300); an all-reduce is enqueued and then waited for with
// hipStreamQuery in a loop that polls ncclCommGetAsyncError (a dead peer) under a wall-clock bound far above any rank-to-rank skew (6 h; QEMB_COMM_TIMEOUT_S).  On a timeout or an asynchronous RCCL error the communicator is
// aborted (ncclCommAbort, from a detached thread: it may itself block on a wedged kernel), the call returns QEMB_ERR_DEVICE and every later
// call fails at once -- the rank is expected to exit non-zero, which is what makes its launcher stop the others.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include "dev_ops.h"
#include "hip_common.h"

namespace qemb {
namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                        // optional (older RCCL): without it a broken communicator is leaked
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr; // optional
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;
std::mutex g_comm_mutex;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
bool g_broken = false;            // a collective timed out or failed: the communicator was aborted, every later call fails
double* g_stage = nullptr;        // device staging buffer of the all-reduce (grown on demand)
size_t g_stage_elems = 0;
double* g_pinned = nullptr;       // pinned host image of it: both copies of a call are asynchronous on the stream
size_t g_pinned_elems = 0;

int load_rccl() {
  if (g_api.handle) return QEMB_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  std::string tried;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    tried += std::string(tried.empty() ? "" : "; ") + dlerror();
  }
  if (!h) { set_error("qemb_comm: RCCL is not available (" + tried + ")"); return QEMB_ERR_DEVICE; }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");
  a.CommGetAsyncError = (decltype(a.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString) {
    set_error("qemb_comm: librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString");
    return QEMB_ERR_DEVICE;
  }
  g_api = a;
  return QEMB_OK;
}
#define RCCL_TRY(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess) {                                                                                   \
      set_error(std::string(#expr) + " failed: " + g_api.GetErrorString(_r) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
      return QEMB_ERR_DEVICE;                                                                                  \
    }                                                                                                          \
  } while (0)
// Wall-clock bound of ONE all-reduce, timed from when THIS rank enqueues its collective -- so it also counts the time the slowest rank still needs to reach
// the exchange (an LPT-partitioned sweep of heterogeneous fragments, a cold first sweep).  It is therefore not a peer-death detector: a dead peer shows as an
// asynchronous RCCL error (polled below) and as a child exit in the launcher, which stops the other ranks.  Default: 6 hours, far above any sweep skew;
// QEMB_COMM_TIMEOUT_S=<seconds> tightens it (tests), QEMB_COMM_TIMEOUT_S=0 removes it.
constexpr double COMM_TIMEOUT_DEFAULT_S = 21600.0;
double comm_timeout_s() {
  const char* e = std::getenv("QEMB_COMM_TIMEOUT_S");
  if (!e) return COMM_TIMEOUT_DEFAULT_S;
  const double v = std::atof(e);
  return v > 0 ? v : 0.0;      // 0: no wall-clock bound
}
// the rendezvous waits for processes that are still starting (a cold container pages the image in for minutes, and not at the same pace for
// every rank): its own bound, QEMB_COMM_INIT_TIMEOUT_S; unset, an explicit QEMB_COMM_TIMEOUT_S holds for it too, else 300 s
double comm_init_timeout_s() {
  if (const char* e = std::getenv("QEMB_COMM_INIT_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) return v; }
  if (std::getenv("QEMB_COMM_TIMEOUT_S") && comm_timeout_s() > 0) return comm_timeout_s();
  return 300.0;
}

// the communicator can no longer be used: abort it off-thread (the call may block while a collective kernel spins on a peer that is gone),
// keep the staging buffers (that kernel may still touch them) and make every later call fail at once.  Caller holds g_comm_mutex.
void abandon_comm() {
  ncclComm_t c = g_comm;
  g_comm = nullptr; g_broken = true;
  g_stage = nullptr; g_stage_elems = 0; g_pinned = nullptr; g_pinned_elems = 0;      // leaked on purpose
  if (c && g_api.CommAbort) {
    auto abort_fn = g_api.CommAbort;
    std::thread([c, abort_fn] { (void)abort_fn(c); }).detach();
  }
}
}  // namespace

static_assert(COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "qemb_comm id size follows ncclUniqueId");

int dev_comm_unique_id(void* id128) {
  if (!id128) { set_error("qemb_comm_unique_id: null buffer"); return QEMB_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  RCCL_TRY(g_api.GetUniqueId(&id));
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return QEMB_OK;
}

int dev_comm_init(int rank, int world, const void* id128) {
  if (world < 1 || rank < 0 || rank >= world || !id128) { set_error("qemb_comm_init: need 0 <= rank < world and an id"); return QEMB_ERR_ARG; }
  if (!hip_stream()) { set_error("qemb_comm_init: call qemb_init(device) first (the communicator is bound to the library's device and stream)"); return QEMB_ERR_DEVICE; }
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (g_comm) { set_error("qemb_comm_init: a communicator already exists (one per process; qemb_comm_destroy first)"); return QEMB_ERR_ARG; }
  if (int rc = load_rccl()) return rc;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  // ncclCommInitRank is collective -- it returns when every rank has joined -- so it runs in a helper thread and this one waits with a bound
  struct Rendezvous {
    std::mutex m; std::condition_variable cv; bool done = false;
    ncclResult_t r = ncclSuccess; hipError_t he = hipSuccess; ncclComm_t c = nullptr;
  };
  auto st = std::make_shared<Rendezvous>();
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  auto init_fn = g_api.CommInitRank;
  std::thread([st, id, world, rank, dev, init_fn] {
    ncclComm_t c = nullptr;
    hipError_t he = hipSetDevice(dev);
    ncclResult_t r = (he == hipSuccess) ? init_fn(&c, world, id, rank) : ncclSuccess;
    std::lock_guard<std::mutex> lk(st->m);
    st->c = c; st->r = r; st->he = he; st->done = true;
    st->cv.notify_all();
  }).detach();
  const double lim = comm_init_timeout_s();
  {
    std::unique_lock<std::mutex> lk(st->m);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(lim), [&] { return st->done; })) {
      g_broken = true;          // the helper is still inside RCCL; nothing of it is touched again
      set_error("qemb_comm_init: rank " + std::to_string(rank) + " of " + std::to_string(world) + " waited " + std::to_string((int)lim) +
                " s for the other ranks to join (QEMB_COMM_INIT_TIMEOUT_S / QEMB_COMM_TIMEOUT_S); a rank never started or is gone");
      return QEMB_ERR_DEVICE;
    }
    if (st->he != hipSuccess) { set_error(std::string("qemb_comm_init: hipSetDevice failed: ") + hipGetErrorString(st->he)); return QEMB_ERR_DEVICE; }
    if (st->r != ncclSuccess) { set_error(std::string("ncclCommInitRank failed: ") + g_api.GetErrorString(st->r)); return QEMB_ERR_DEVICE; }
    g_comm = st->c;
  }
  g_rank = rank; g_world = world; g_broken = false;
  return QEMB_OK;
}

int dev_comm_info(int* rank, int* world) {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (rank) *rank = g_comm ? g_rank : 0;
  if (world) *world = g_comm ? g_world : 1;
  return QEMB_OK;
}

int dev_comm_allreduce(double* host_buf, int64_t n, int op) {
  if (n < 0 || (n > 0 && !host_buf) || (op != COMM_SUM && op != COMM_MAX)) { set_error("qemb_comm_allreduce: bad arguments"); return QEMB_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (g_broken) { set_error("qemb_comm_allreduce: the communicator was aborted after a failed or timed-out collective"); return QEMB_ERR_DEVICE; }
  if (!g_comm) { set_error("qemb_comm_allreduce: no communicator (qemb_comm_init)"); return QEMB_ERR_DEVICE; }
  if (n == 0) return QEMB_OK;
  hipStream_t s = hip_stream();
  if ((size_t)n > g_stage_elems) {
    const size_t want = ((size_t)n + 1023) / 1024 * 1024;
    if (g_stage) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(g_stage)); g_stage = nullptr; g_stage_elems = 0; }
    if (g_pinned) { HIP_TRY(hipHostFree(g_pinned)); g_pinned = nullptr; g_pinned_elems = 0; }
    HIP_TRY(hipMalloc((void**)&g_stage, want * sizeof(double)));
    g_stage_elems = want;
    HIP_TRY(hipHostMalloc((void**)&g_pinned, want * sizeof(double), hipHostMallocDefault));
    g_pinned_elems = want;
  }
  memcpy(g_pinned, host_buf, (size_t)n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(g_stage, g_pinned, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  RCCL_TRY(g_api.AllReduce(g_stage, g_stage, (size_t)n, ncclDouble, op == COMM_SUM ? ncclSum : ncclMax, g_comm, s));
  HIP_TRY(hipMemcpyAsync(g_pinned, g_stage, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  // bounded wait: a peer that died leaves the collective kernel spinning for ever, so never hipStreamSynchronize here
  const double lim = comm_timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 1;; ++spins) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      abandon_comm();
      set_error(std::string("qemb_comm_allreduce: the stream failed while the collective was in flight: ") + hipGetErrorString(q));
      return QEMB_ERR_DEVICE;
    }
    if ((spins & 63) == 0) {
      if (g_api.CommGetAsyncError) {
        ncclResult_t ar = ncclSuccess;
        if (g_api.CommGetAsyncError(g_comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
          const std::string why = g_api.GetErrorString(ar);
          abandon_comm();
          set_error("qemb_comm_allreduce: RCCL reported an asynchronous error (" + why + "); the communicator was aborted");
          return QEMB_ERR_DEVICE;
        }
      }
      if (lim > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
        abandon_comm();
        set_error("qemb_comm_allreduce: rank " + std::to_string(g_rank) + " of " + std::to_string(g_world) + " waited " + std::to_string((int)lim) +
                  " s for the all-reduce (QEMB_COMM_TIMEOUT_S); a rank is gone -- the communicator was aborted");
        return QEMB_ERR_DEVICE;
      }
      if (spins > 20000) std::this_thread::sleep_for(std::chrono::microseconds(50));     // a few ms of pure polling first: the exchange is latency bound
    }
  }
  memcpy(host_buf, g_pinned, (size_t)n * sizeof(double));
  return QEMB_OK;
}

int dev_comm_destroy() {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (!g_comm) { g_broken = false; return QEMB_OK; }      // an abandoned communicator has nothing left to release
  if (hip_stream()) HIP_TRY(hipStreamSynchronize(hip_stream()));
  ncclComm_t c = g_comm;
  g_comm = nullptr; g_rank = 0; g_world = 1;
  if (g_stage) { (void)hipFree(g_stage); g_stage = nullptr; g_stage_elems = 0; }
  if (g_pinned) { (void)hipHostFree(g_pinned); g_pinned = nullptr; g_pinned_elems = 0; }
  RCCL_TRY(g_api.CommDestroy(c));
  return QEMB_OK;
}

}  // namespace qemb
