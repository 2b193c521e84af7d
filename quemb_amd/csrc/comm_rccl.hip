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
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include <string>
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
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;
std::mutex g_comm_mutex;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
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
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  RCCL_TRY(g_api.CommInitRank(&c, world, id, rank));     // collective: returns when every rank has joined
  g_comm = c; g_rank = rank; g_world = world;
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
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(host_buf, g_pinned, (size_t)n * sizeof(double));
  return QEMB_OK;
}

int dev_comm_destroy() {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (!g_comm) return QEMB_OK;
  if (hip_stream()) HIP_TRY(hipStreamSynchronize(hip_stream()));
  ncclComm_t c = g_comm;
  g_comm = nullptr; g_rank = 0; g_world = 1;
  if (g_stage) { (void)hipFree(g_stage); g_stage = nullptr; g_stage_elems = 0; }
  if (g_pinned) { (void)hipHostFree(g_pinned); g_pinned = nullptr; g_pinned_elems = 0; }
  RCCL_TRY(g_api.CommDestroy(c));
  return QEMB_OK;
}

}  // namespace qemb
