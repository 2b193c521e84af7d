// dev_ops_hip.hip -- gfx950 implementation of dev_ops.h (everything except the MFMA GEMM, which lives
// in gemm_f64.hip, and the Jacobi / Cholesky kernels in linalg_f64.hip).
//
// These are the HBM-bound pieces of the hot path: tensor permutations between the GEMM-shaped CCSD
// contractions, the tau / denominator / DIIS vector work (reference: PySCF ccsd.update_amps as driven by
// molbe/solver.py:907), the J/K contractions of the fragment RHF (molbe/helper.py:64 dot_eri_dm) and the
// packed-pair (s4/s8) <-> full index transforms (ao2mo.restore at helper.py:189, mbe.py:1155).
// They are written for coalescing (16-byte lanes where the layout allows, LDS-tiled transposes when the
// contiguous dimension changes) rather than reshaped into GEMMs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <atomic>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include "dev_ops.h"
#include "hip_common.h"
#include "grouped_launch.h"

namespace qemb {
// XCD-aware logical block index of the tiled HBM passes (round 4; profiles/r04_hbm_pmc.json).  Workgroups are dealt round-robin over the eight
// XCDs, each with its own L2, so neighbouring 32 x 32 tiles -- whose 256-byte row pieces start at arbitrary offsets and share their first and
// last 128-byte lines -- ran on different XCDs and every shared line was fetched from HBM twice: FETCH_SIZE showed 1.2-1.5 x the algorithmic
// reads for the unpack, scatter and finishing passes.  Here the linear block number is mapped so that each XCD works through ONE contiguous
// range of the logical order (x fastest): neighbours in x run on the same XCD at about the same time and meet in its L2.  A bijection of the
// grid: placement is a matter of speed only (inside a grouped launch the member's blocks are offset and it is merely another permutation).
__device__ __forceinline__ uint3 xcd_logical_block(const uint3 BID, const uint3 GDIM) {
  const unsigned long long total = (unsigned long long)GDIM.x * GDIM.y * GDIM.z;
  const unsigned long long lin = ((unsigned long long)BID.z * GDIM.y + BID.y) * GDIM.x + BID.x;
  const unsigned long long q = total >> 3, r = total & 7ull;
  const unsigned long long xcd = lin & 7ull, k = lin >> 3;
  const unsigned long long base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const unsigned long long L = base + k;
  const unsigned long long xy = (unsigned long long)GDIM.x * GDIM.y;
  return make_uint3((unsigned)(L % GDIM.x), (unsigned)((L % xy) / GDIM.x), (unsigned)(L / xy));
}


// ------------------------------------------------------------------------------------------------
// error channel + device state
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

// One EXECUTION CONTEXT = one HIP stream with its own reduction scratch, workspaces, block cache, timers and capture state.
// Every host thread is bound to a context (the process default unless dev_ctx_bind is called), so several fragments can be
// driven concurrently from several host threads, each on its own stream: fragments whose kernels are latency bound
// (n ~ 40-60) then overlap on the device.  Nothing is shared between contexts, so no locking is needed on the hot path.
// Device-time laps: every begin/end pair of a slot is one (start, stop) event pair.  Pairs are recycled through a per-slot free list
// and collected without a host sync as soon as their stop event has completed, so a long optimisation that never reads its timers
// keeps a bounded number of live events (TIMER_MAX_PENDING per slot); a begin whose end was never recorded (early return of the
// bracketed region) is dropped at the next begin / collect instead of poisoning the slot.  QEMB_TIMERS=0 turns the laps off.
struct TimerLap { hipEvent_t e0, e1; bool ended; };
struct TimerSlot { double total_ms = 0; int64_t count = 0; std::vector<TimerLap> pending; std::vector<std::pair<hipEvent_t, hipEvent_t>> spare; };
static constexpr size_t TIMER_MAX_PENDING = 64;
static constexpr int NPART = 2048;
struct DevCtx {
  hipStream_t stream = nullptr;
  bool capturing = false;            // inside hipStreamBeginCapture .. EndCapture (dev_graph_*)
  double* partials = nullptr;        // reduction scratch (8 x NPART doubles: up to eight dot products per pass)
  double* ws = nullptr;              // growable workspace for contract_mid / k_pairs partials
  size_t ws_bytes = 0;
  double* gws = nullptr;             // split-K partial-sum workspace of the GEMM
  size_t gws_bytes = 0;
  // parallel regions of a tape capture (dev_region_*): the marks that delimit regions / chains in the captured chain of nodes, and a bump offset so that
  // every split-K product recorded inside a region gets its own slice of gws (the chains of a region may run side by side)
  std::vector<std::pair<hipGraphNode_t, int>> marks;      // (last captured node at the time of the mark, mark)
  bool tape_capture = false, in_region = false;
  bool alloc_trace = false; unsigned long long alloc_hash = 0;      // dev_alloc_trace_*
  size_t gws_bump = 0;
  TimerSlot timers[TIMER_NSLOTS];
  std::map<size_t, std::vector<void*>> pool;   // caching allocator (see below)
  size_t pool_bytes = 0;
  // pinned staging of small host <-> device copies (dev_h2d / dev_d2h): a ring of slots, each with the event of its last upload
  unsigned char* stage = nullptr;
  hipEvent_t stage_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool stage_busy[8] = {false, false, false, false, false, false, false, false};
  int stage_next = 0;
};
struct LiveBlock { size_t bytes; DevCtx* owner; };
static std::map<void*, LiveBlock> g_live_blocks;   // every block handed out by dev_alloc, any context
static std::mutex g_alloc_mutex;                   // allocator bookkeeping only (host side, off the kernel path)
static int g_device = -1;
static DevCtx g_default_ctx;
static std::vector<DevCtx*> g_extra_ctx;          // created by dev_ctx_count(n); index k >= 1
static std::mutex g_ctx_mutex;
static int g_ctx_parts = [] { const char* e = std::getenv("QEMB_CU_SPLIT"); return e ? std::atoi(e) : 0; }();      // dev_ctx_partition
static thread_local DevCtx* t_ctx = nullptr;
static inline DevCtx& ctx() { return t_ctx ? *t_ctx : g_default_ctx; }
#define g_stream (ctx().stream)
#define g_capturing (ctx().capturing)
#define g_partials (ctx().partials)
#define g_ws (ctx().ws)
#define g_ws_bytes (ctx().ws_bytes)
#define g_gws (ctx().gws)
#define g_gws_bytes (ctx().gws_bytes)
#define g_timers (ctx().timers)
#define g_pool (ctx().pool)
#define g_pool_bytes (ctx().pool_bytes)

hipStream_t hip_stream() { return g_stream; }
const char* dev_backend_name() { return "hip-gfx950"; }

int dev_init(int device) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { set_error("no HIP device visible: libqemb_hip has no CPU fallback"); return QEMB_ERR_DEVICE; }
  if (device < 0 || device >= ndev) { set_error("dev_init: device index out of range"); return QEMB_ERR_ARG; }
  DevCtx& c = g_default_ctx;
  if (c.stream && g_device == device) return QEMB_OK;
  if (c.stream) {
    // One process drives one GPU (one rank per GPU): device memory, streams and cached blocks of the first device are live behind
    // handles the caller still holds, so a second device is refused rather than half torn down.
    set_error("libqemb_hip is already initialised on device " + std::to_string(g_device) + "; one process drives one GPU");
    return QEMB_ERR_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void**)&c.partials, (8 * NPART + 8) * sizeof(double)));
  HIP_TRY(hipMemset(c.partials + 8 * NPART, 0, 8 * sizeof(double)));      // the "workgroups done" counter of the single-launch reductions
  g_device = device;
  return QEMB_OK;
}
// The stream of extra context j (0-based).  With a CU partition (dev_ctx_partition) it is created on the compute units whose index is congruent to j mod parts,
// so fragments in flight run on disjoint parts of the chip: the big GEMM workgroups fill a CU's registers, so two fragments whose kernels alternate on ALL CUs
// only overlap in the gaps, while on disjoint halves one fragment's HBM-bound passes run beside the other's MFMA-bound products (n = 220, four in flight:
// +0.7-2 % by box; contiguous halves measured worse than interleaved ones, three or four parts worse than two).  A runtime that refuses the mask gets a plain stream.
static hipError_t create_ctx_stream(hipStream_t* stream, int j) {
  hipError_t e = hipErrorUnknown;
  // (under rocprofv3 the process crashed in its exit handlers when CU-masked streams existed -- and hung when they were destroyed from an atexit handler --, so a
  //  profiled run keeps plain streams, as it keeps eager launches instead of graph replay)
  static const bool profiled = [] {
    const char* pre = std::getenv("LD_PRELOAD");
    return std::getenv("ROCP_TOOL_LIBRARIES") != nullptr || (pre && std::strstr(pre, "rocprofiler"));
  }();
  if (g_ctx_parts >= 2 && !profiled) {
    hipDeviceProp_t prop;
    int ncu = 256;
    if (hipGetDeviceProperties(&prop, g_device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int cu = 0; cu < ncu; ++cu) if (cu % g_ctx_parts == j % g_ctx_parts) mask[(size_t)cu >> 5] |= 1u << (cu & 31);
    e = hipExtStreamCreateWithCUMask(stream, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { (void)hipGetLastError(); *stream = nullptr; }
  }
  if (e != hipSuccess) e = hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
  return e;
}
// make sure contexts 0..n-1 exist (0 = the default one); returns the number available
int dev_ctx_count(int n) {
  if (!g_default_ctx.stream) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  HIP_TRY(hipSetDevice(g_device));
  while ((int)g_extra_ctx.size() + 1 < n) {
    DevCtx* c = new DevCtx();
    hipError_t e = create_ctx_stream(&c->stream, (int)g_extra_ctx.size());
    if (e == hipSuccess) e = hipMalloc((void**)&c->partials, (8 * NPART + 8) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(c->partials + 8 * NPART, 0, 8 * sizeof(double));
    if (e != hipSuccess) { delete c; set_error(std::string("dev_ctx_count: ") + hipGetErrorString(e)); return QEMB_ERR_DEVICE; }
    g_extra_ctx.push_back(c);
  }
  return (int)g_extra_ctx.size() + 1;
}
// Contexts 1, 2, ... are spread over `parts` disjoint, interleaved sets of compute units (0 / 1: the whole chip each).  Existing extra contexts are drained and
// get a new stream (their workspaces and cached blocks stay: everything on the old stream has finished); call it between sweeps, not while threads are solving.
int dev_ctx_partition(int parts) {
  if (parts < 0 || parts > 8) { set_error("dev_ctx_partition: 0 <= parts <= 8"); return QEMB_ERR_ARG; }
  if (!g_default_ctx.stream) { g_ctx_parts = parts; return QEMB_OK; }
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  if (parts == g_ctx_parts || (parts <= 1 && g_ctx_parts <= 1)) { g_ctx_parts = parts; return QEMB_OK; }
  g_ctx_parts = parts;
  HIP_TRY(hipSetDevice(g_device));
  for (size_t j = 0; j < g_extra_ctx.size(); ++j) {
    DevCtx* c = g_extra_ctx[j];
    if (c->capturing) { set_error("dev_ctx_partition: a context is capturing"); return QEMB_ERR_DEVICE; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (event pairs of the timers were recorded on the old stream and have completed; they stay valid)
    HIP_TRY(hipStreamDestroy(c->stream));
    c->stream = nullptr;
    HIP_TRY(create_ctx_stream(&c->stream, (int)j));
  }
  return QEMB_OK;
}
// bind the calling host thread to context k (HIP's current device is per thread as well)
int dev_ctx_bind(int k) {
  if (!g_default_ctx.stream) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  if (k < 0 || k > (int)g_extra_ctx.size()) { set_error("dev_ctx_bind: no such context (call dev_ctx_count first)"); return QEMB_ERR_ARG; }
  HIP_TRY(hipSetDevice(g_device));
  t_ctx = (k == 0) ? &g_default_ctx : g_extra_ctx[k - 1];
  return QEMB_OK;
}
#define REQUIRE_INIT()                                                                          \
  do { if (!g_stream) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; } } while (0)

int dev_sync() { REQUIRE_INIT(); HIP_TRY(hipStreamSynchronize(g_stream)); return QEMB_OK; }
int dev_sync_device() { REQUIRE_INIT(); HIP_TRY(hipDeviceSynchronize()); return QEMB_OK; }
// ---- caching allocator --------------------------------------------------------------------------------------
// Every fragment solve allocates the same handful of multi-GB buffers (two n^4 ping-pong tensors, the v^4 ladder
// operand, ...).  hipMalloc / hipFree of such blocks cost ~0.1 s each and hipFree synchronises the device, so
// freed blocks are parked in an exact-size free list and handed back to the next request of that size.  All work
// is on ONE stream, so reuse is stream-ordered and needs no synchronisation.  dev_trim() / an allocation failure
// releases the parked blocks.

static std::atomic<long long> g_alloc_misses{0}, g_alloc_miss_ns{0}, g_alloc_miss_bytes{0}, g_driver_frees{0}, g_driver_free_ns{0};
static hipError_t timed_hip_free(void* q) {      // every hipFree of the pool goes through here (it waits for the device: worth knowing when it happens)
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipFree(q);
  g_driver_frees += 1; g_driver_free_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return e;
}
static void trim_ctx_locked(DevCtx& c) {      // g_alloc_mutex held; the context's stream has been drained
  for (auto& kv : c.pool) for (void* q : kv.second) (void)timed_hip_free(q);
  c.pool.clear(); c.pool_bytes = 0;
}
static int trim_all_contexts();
int dev_trim_all() { REQUIRE_INIT(); return trim_all_contexts(); }
int dev_trim() {
  if (!g_stream) return QEMB_OK;
  HIP_TRY(hipStreamSynchronize(g_stream));
  std::lock_guard<std::mutex> lock(g_alloc_mutex);
  trim_ctx_locked(ctx());
  return QEMB_OK;
}
// An allocation failed: hand the parked blocks of EVERY context back to the driver (a multi-GB block cached by another stream's
// context is as good as free).  Each context's stream is drained first, so none of its parked blocks is still in use.
static int trim_all_contexts() {
  std::vector<DevCtx*> all;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    all.push_back(&g_default_ctx);
    for (DevCtx* c : g_extra_ctx) all.push_back(c);
  }
  for (DevCtx* c : all) if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
  std::lock_guard<std::mutex> lock(g_alloc_mutex);
  for (DevCtx* c : all) trim_ctx_locked(*c);
  return QEMB_OK;
}
// Cached blocks of one context are capped (QEMB_POOL_CAP_GB; default: half of the device's memory, 144 GB on an MI355X -- the working
// set of an n = 300 fragment solve, 2 x 32.5 GB of transform buffers + the 21 GB ladder operands, stays parked between solves): beyond it a
// released block goes back to the driver.  An allocation that fails trims every context's parked blocks and retries (dev_alloc).
static size_t pool_cap_bytes() {
  static const size_t cap = [] {
    const char* e = std::getenv("QEMB_POOL_CAP_GB");
    if (e) return (size_t)(std::atof(e) * (double)(1ull << 30));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return (size_t)96 << 30;
    return total_b / 2;
  }();
  return cap;
}
int dev_alloc_stats(long long* n, long long* nfree, double* ms, double* gb, int reset) {
  if (n) *n = g_alloc_misses.load();
  if (nfree) *nfree = g_driver_frees.load();
  if (ms) *ms = (double)(g_alloc_miss_ns.load() + g_driver_free_ns.load()) * 1e-6;
  if (gb) *gb = (double)g_alloc_miss_bytes.load() * 1e-9;
  if (reset) { g_alloc_misses = 0; g_alloc_miss_ns = 0; g_alloc_miss_bytes = 0; g_driver_frees = 0; g_driver_free_ns = 0; }
  return QEMB_OK;
}
static inline unsigned long long hash_mix(unsigned long long h, unsigned long long x) {      // (splitmix64 step over h ^ x)
  unsigned long long z = (h ^ x) + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
int dev_alloc(void** p, size_t bytes) {
  REQUIRE_INIT();
  if (bytes == 0) bytes = 16;
  bytes = (bytes + 255) / 256 * 256;
  {
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    auto it = g_pool.find(bytes);
    if (it != g_pool.end() && !it->second.empty()) {
      *p = it->second.back(); it->second.pop_back(); g_pool_bytes -= bytes;
      g_live_blocks[*p] = LiveBlock{bytes, &ctx()};
      if (ctx().alloc_trace) ctx().alloc_hash = hash_mix(hash_mix(ctx().alloc_hash, (unsigned long long)bytes), (unsigned long long)reinterpret_cast<uintptr_t>(*p));
      return QEMB_OK;
    }
  }
  if (g_capturing) { set_error("device allocation inside a captured region"); return QEMB_ERR_ALLOC; }   // pooled blocks are fine, a real hipMalloc is not
  struct MissTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); size_t b;
    ~MissTimer() { g_alloc_misses += 1; g_alloc_miss_bytes += (long long)b; g_alloc_miss_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
  } miss_timer;
  miss_timer.b = bytes;
  {
    // Do not walk INTO an out-of-memory condition: on this runtime a hipMalloc beyond what is free has been seen to crash inside the HSA allocator (or to hang)
    // instead of failing (round 5: bench.py's size sweep after the other sections, ~200 GB parked in the pools of seven contexts).  When the request does not fit
    // what the driver reports free, the parked blocks of every context go back first.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes + ((size_t)512 << 20) > free_b) {
      const int rc = trim_all_contexts();
      if (rc) return rc;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    int rc = trim_all_contexts();
    if (rc) return rc;
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) { set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e)); return QEMB_ERR_ALLOC; }
  std::lock_guard<std::mutex> lock(g_alloc_mutex);
  g_live_blocks[*p] = LiveBlock{bytes, &ctx()};
  if (ctx().alloc_trace) ctx().alloc_hash = hash_mix(hash_mix(ctx().alloc_hash, (unsigned long long)bytes), (unsigned long long)reinterpret_cast<uintptr_t>(*p));
  return QEMB_OK;
}
void dev_alloc_trace_begin() { ctx().alloc_trace = true; ctx().alloc_hash = 0x9e3779b97f4a7c15ull; }
unsigned long long dev_alloc_trace_end() {
  ctx().alloc_trace = false;
  unsigned long long h = ctx().alloc_hash;
  h = hash_mix(h, (unsigned long long)reinterpret_cast<uintptr_t>(ctx().partials));
  h = hash_mix(h, (unsigned long long)reinterpret_cast<uintptr_t>(ctx().ws));
  h = hash_mix(h, (unsigned long long)reinterpret_cast<uintptr_t>(ctx().gws));
  return hash_mix(h, (unsigned long long)reinterpret_cast<uintptr_t>(&ctx()));
}
int dev_free(void* p) {
  if (!p) return QEMB_OK;
  LiveBlock blk{0, nullptr};
  {
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    auto it = g_live_blocks.find(p);
    if (it != g_live_blocks.end()) { blk = it->second; g_live_blocks.erase(it); }
  }
  if (!blk.owner) { HIP_TRY(hipStreamSynchronize(g_stream)); HIP_TRY(timed_hip_free(p)); return QEMB_OK; }
  // a block goes back to the cache of the context that allocated it; when another context releases it, that context's
  // stream is drained first so that the owner cannot reuse the block under kernels still in flight
  if (blk.owner != &ctx() && g_stream) HIP_TRY(hipStreamSynchronize(g_stream));
  {
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    if (blk.owner->pool_bytes + blk.bytes <= pool_cap_bytes()) {
      // kept sorted by descending address: dev_alloc takes the back, the LOWEST address of the size class.  Which block a request gets then depends on the set of
      // parked blocks only, not on the order they came back in -- a solve that allocates what the last one did finds its buffers at the same addresses (with a
      // stack the buffers of one size class traded places from solve to solve), which is what lets a fragment keep its recorded update (dev_alloc_trace_*)
      std::vector<void*>& parked = blk.owner->pool[blk.bytes];
      parked.insert(std::upper_bound(parked.begin(), parked.end(), p, std::greater<void*>()), p);
      blk.owner->pool_bytes += blk.bytes;
      return QEMB_OK;
    }
  }
  HIP_TRY(hipStreamSynchronize(g_stream));     // over the cap: really free it (hipFree also waits for the device)
  HIP_TRY(timed_hip_free(p));
  return QEMB_OK;
}
// Small copies between PAGEABLE host memory and the device go through pinned slots of the calling context (round 5).  The runtime stages a pageable copy itself,
// synchronously and under locks that the host threads of a batched sweep queue on (six fragments x ~8 small copies in the RHF phase of an octane sweep: the phase took
// 3 x the time of one fragment alone).  Upload: the bytes are copied into the next slot of a ring and leave from there asynchronously -- no wait at all (the caller's
// buffer is free on return, the stream orders the copy before later launches; a slot is reused only after the event of its last upload).  Download: device -> slot,
// one stream wait, slot -> destination.  QEMB_STAGED_COPIES=0: the plain calls, for A/B runs.
static constexpr size_t STAGE_SLOT = 256 * 1024;      // (n x n doubles up to n = 181)
static bool stage_ready(DevCtx& c) {
  static const bool on = !(std::getenv("QEMB_STAGED_COPIES") && std::atoi(std::getenv("QEMB_STAGED_COPIES")) == 0);
  if (!on || c.capturing) return false;
  if (c.stage) return true;
  void* q = nullptr;
  if (hipHostMalloc(&q, 8 * STAGE_SLOT, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
  for (int k = 0; k < 8; ++k) if (hipEventCreateWithFlags(&c.stage_ev[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(q); return false; }
  c.stage = (unsigned char*)q;
  return true;
}
static int h2d_impl(void* dst, const void* src, size_t bytes, bool wait);
// dev_h2d returns when the bytes are on the device (any stream may read them); dev_h2d_async only orders the copy on the calling context's stream -- for data that
// the same context consumes (the source buffer is free on return either way)
int dev_h2d(void* dst, const void* src, size_t bytes) { return h2d_impl(dst, src, bytes, true); }
int dev_h2d_async(void* dst, const void* src, size_t bytes) { return h2d_impl(dst, src, bytes, false); }
static int h2d_impl(void* dst, const void* src, size_t bytes, bool wait) {
  REQUIRE_INIT();
  DevCtx& c = ctx();
  if (bytes > 0 && bytes <= STAGE_SLOT && stage_ready(c)) {
    const int k = c.stage_next; c.stage_next = (k + 1) & 7;
    if (c.stage_busy[k]) { HIP_TRY(hipEventSynchronize(c.stage_ev[k])); c.stage_busy[k] = false; }
    unsigned char* slot = c.stage + (size_t)k * STAGE_SLOT;
    std::memcpy(slot, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c.stream));
    if (wait) { HIP_TRY(hipStreamSynchronize(c.stream)); return QEMB_OK; }
    HIP_TRY(hipEventRecord(c.stage_ev[k], c.stream));
    c.stage_busy[k] = true;
    return QEMB_OK;
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));   // pageable host memory: keep the contract simple
  return QEMB_OK;
}
int dev_d2h(void* dst, const void* src, size_t bytes) {
  REQUIRE_INIT();
  DevCtx& c = ctx();
  if (bytes > 0 && bytes <= STAGE_SLOT && stage_ready(c)) {
    const int k = c.stage_next; c.stage_next = (k + 1) & 7;
    if (c.stage_busy[k]) { HIP_TRY(hipEventSynchronize(c.stage_ev[k])); c.stage_busy[k] = false; }
    unsigned char* slot = c.stage + (size_t)k * STAGE_SLOT;
    HIP_TRY(hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    std::memcpy(dst, slot, bytes);
    return QEMB_OK;
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return QEMB_OK;
}
int dev_d2h_async(void* dst, const void* src, size_t bytes) {
  REQUIRE_INIT();
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream));
  return QEMB_OK;
}
// Pinned host blocks are small (a few scalars on their way back from the device) and short lived (one per DIIS object and solver):
// hipHostMalloc / hipHostFree cost tens of microseconds each and the free waits for the device, so released blocks are parked by size
// class and handed out again (process wide; never returned to the driver).
static std::mutex g_pinned_mutex;
static std::map<size_t, std::vector<void*>> g_pinned_free;
static std::map<void*, size_t> g_pinned_live;
int dev_pinned_alloc(void** p, size_t bytes) {
  REQUIRE_INIT();
  const size_t cls = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned_free.find(cls);
    if (it != g_pinned_free.end() && !it->second.empty()) { *p = it->second.back(); it->second.pop_back(); g_pinned_live[*p] = cls; return QEMB_OK; }
  }
  HIP_TRY(hipHostMalloc(p, cls, hipHostMallocDefault));
  {  // kernels write their scalar results straight into these blocks (single-launch reductions): the host address must be the device address
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, *p, 0));
    if (dp != *p) { (void)hipHostFree(*p); *p = nullptr; set_error("pinned host memory is not mapped at its host address on this platform"); return QEMB_ERR_DEVICE; }
  }
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  g_pinned_live[*p] = cls;
  return QEMB_OK;
}
int dev_pinned_free(void* p) {
  if (!p) return QEMB_OK;
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  auto it = g_pinned_live.find(p);
  if (it == g_pinned_live.end()) return QEMB_OK;
  g_pinned_free[it->second].push_back(p);
  g_pinned_live.erase(it);
  return QEMB_OK;
}
// (inside a capture a device-to-device copy is issued as a KERNEL: a captured hipMemcpyAsync becomes a memcpy node whose parameters this runtime hands back in a
//  form neither hipMemcpy3DAsync nor a linear replay accepts, and tapes replay node by node)
__global__ void __launch_bounds__(256) copy_words_kernel(double* __restrict__ dst, const double* __restrict__ src, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
int dev_d2d(void* dst, const void* src, size_t bytes) {
  REQUIRE_INIT();
  if (g_capturing && bytes % 8 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 7) == 0) {
    const long long n = (long long)(bytes / 8);
    if (n > 0) hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, g_stream, (double*)dst, (const double*)src, n);
    HIP_TRY(hipGetLastError());
    return QEMB_OK;
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
  return QEMB_OK;
}
int dev_mem_info(size_t* free_b, size_t* total_b) { REQUIRE_INIT(); HIP_TRY(hipMemGetInfo(free_b, total_b)); return QEMB_OK; }

double* gemm_workspace(size_t bytes) {
  if (ctx().in_region) {      // recorded inside a parallel region: a slice of its own (the region's chains do not share scratch)
    const size_t need = (bytes + 255) / 256 * 256;
    if (!g_gws || ctx().gws_bump + need > g_gws_bytes) { set_error("split-K workspace too small for a parallel region"); return nullptr; }
    double* p = (double*)((unsigned char*)g_gws + ctx().gws_bump);
    ctx().gws_bump += need;
    return p;
  }
  if (bytes <= g_gws_bytes) return g_gws;
  if (g_capturing) { set_error("split-K workspace growth inside a captured region"); return nullptr; }
  if (g_gws) { (void)hipStreamSynchronize(g_stream); (void)timed_hip_free(g_gws); g_gws = nullptr; g_gws_bytes = 0; }
  const size_t want = bytes < ((size_t)64 << 20) ? ((size_t)64 << 20) : bytes;
  if (hipMalloc((void**)&g_gws, want) != hipSuccess) { set_error("split-K workspace hipMalloc failed"); return nullptr; }
  g_gws_bytes = want;
  return g_gws;
}

static int ensure_ws(size_t bytes) {
  if (bytes <= g_ws_bytes) return QEMB_OK;
  if (g_capturing) { set_error("workspace growth inside a captured region"); return QEMB_ERR_ALLOC; }
  if (g_ws) { HIP_TRY(hipStreamSynchronize(g_stream)); HIP_TRY(timed_hip_free(g_ws)); g_ws = nullptr; g_ws_bytes = 0; }
  hipError_t e = hipMalloc((void**)&g_ws, bytes);
  if (e != hipSuccess) { set_error("workspace hipMalloc failed"); return QEMB_ERR_ALLOC; }
  g_ws_bytes = bytes;
  return QEMB_OK;
}

// ---- stream capture ---------------------------------------------------------------------------------
bool dev_capturing() { return g_capturing; }
int dev_graph_begin(int for_tape) {
  REQUIRE_INIT();
  if (g_capturing) { set_error("dev_graph_begin: already capturing"); return QEMB_ERR_ARG; }
  if (for_tape && !g_gws) { if (!gemm_workspace(1)) return QEMB_ERR_ALLOC; }      // regions slice the workspace: it must exist before the capture
  HIP_TRY(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
  g_capturing = true;
  ctx().tape_capture = for_tape != 0; ctx().in_region = false; ctx().gws_bump = 0; ctx().marks.clear();
  return QEMB_OK;
}
// Parallel regions (round 5).  Between dev_region_begin and dev_region_end the caller declares CHAINS of operations -- dev_region_chain starts the next one --
// that do not depend on each other (no chain reads what another chain of the region writes, no two write the same place): inside a chain the order of the
// calls is kept, across chains it is free.  Executed eagerly, or captured into an executable graph, the calls simply run in program order and the three
// functions do nothing.  Captured for a TAPE they note which node was the newest of the capture at that moment (hipStreamGetCaptureInfo_v2: no node is added --
// markers as captured one-byte memsets cost 0.7 ms of capture per octane BE2 sweep, more than the merged launches gain), from which dev_tape_end recovers the structure, and
// dev_tape_run issues level k of a region -- the k-th operation of every chain, of every fragment of the run -- together: the same kernel of several chains
// and fragments in ONE grouped launch (up to GROUP_MAX members).  The small-fragment CCSD update is ~46 dependent launches; its data flow is
// ~16 levels deep.  The assertion of independence is the caller's; the lock-step tests compare bit for bit with the one-by-one solves.
enum { MARK_REGION_BEGIN = 0xA1, MARK_CHAIN = 0xA2, MARK_REGION_END = 0xA3 };
static bool regions_enabled() {      // QEMB_TAPE_REGIONS=0: no markers are recorded, the tape is the plain sequence (A/B runs)
  static const bool on = !(std::getenv("QEMB_TAPE_REGIONS") && std::atoi(std::getenv("QEMB_TAPE_REGIONS")) == 0);
  return on;
}
static int region_mark(int value) {
  if (!g_capturing || !ctx().tape_capture || !regions_enabled()) return QEMB_OK;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  HIP_TRY(hipStreamGetCaptureInfo_v2(g_stream, &st, &id, &graph, &deps, &ndeps));
  if (st != hipStreamCaptureStatusActive || ndeps > 1) { set_error("dev_region_*: the capture is not a single chain"); return QEMB_ERR_DEVICE; }
  ctx().marks.emplace_back(ndeps ? deps[0] : nullptr, value);
  return QEMB_OK;
}
int dev_region_begin() { if (g_capturing && ctx().tape_capture && regions_enabled()) { ctx().in_region = true; } { const int rc_ = region_mark(MARK_REGION_BEGIN); if (rc_) return rc_; } return region_mark(MARK_CHAIN); }
int dev_region_chain() { return region_mark(MARK_CHAIN); }
int dev_region_end() { ctx().in_region = false; return region_mark(MARK_REGION_END); }
int dev_graph_end(dev_graph_t* out) {
  REQUIRE_INIT();
  if (!g_capturing) { set_error("dev_graph_end: not capturing"); return QEMB_ERR_ARG; }
  g_capturing = false; ctx().tape_capture = false; ctx().in_region = false;
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamEndCapture(g_stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { set_error(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e)); return QEMB_ERR_DEVICE; }
  *out = exec;
  return QEMB_OK;
}
int dev_graph_launch(dev_graph_t g) { REQUIRE_INIT(); HIP_TRY(hipGraphLaunch((hipGraphExec_t)g, g_stream)); return QEMB_OK; }
int dev_graph_destroy(dev_graph_t g) { if (g) (void)hipGraphExecDestroy((hipGraphExec_t)g); return QEMB_OK; }

// ---- grouped launches --------------------------------------------------------------------------------------------------------------
// A kernel written as `__device__ name_body(BID, GDIM, args...)` + a one-line `__global__ name(args...)` wrapper can also run GROUPED: one
// launch whose grid is the concatenation of the members' grids; a workgroup finds its member, takes that member's arguments from a table
// in device memory and runs the body with its block index inside that member.  Kernels are registered by the address of their wrapper,
// which is what a captured graph node carries.
std::map<const void*, GroupInfo>& groupable() { static std::map<const void*, GroupInfo> m; return m; }
bool group_xcd_mode() { static const bool on = [] { const char* e = std::getenv("QEMB_GROUP_XCD"); return e && e[0] != '0'; }(); return on; }
void register_groupable_gemm();         // gemm_f64.hip
static void register_groupable_kernels();   // end of this file (after the kernels)
static void ensure_groupable_registered() {
  static std::once_flag once;
  std::call_once(once, [] { register_groupable_kernels(); register_groupable_gemm(); });
}

// ---- tapes: captured launch sequences executed together -------------------------------------------------------------------------------
// A tape keeps the captured hipGraph alive (it owns the argument storage of its nodes) and lists the nodes in execution order.
struct TapeNode {
  hipGraphNodeType type;
  hipKernelNodeParams k;        // type == Kernel
  hipMemcpy3DParms cpy;         // type == Memcpy
  hipMemsetParams set;          // type == Memset
  int region = -1, chain = -1;  // inside a parallel region (dev_region_*): which region of the tape, which chain of the region
};
struct Tape {
  hipGraph_t graph = nullptr;
  std::vector<TapeNode> nodes;
};
static thread_local long long t_tape_launches = 0, t_tape_grouped = 0, t_tape_ops = 0;

int dev_tape_end(dev_tape_t* out) {
  REQUIRE_INIT();
  if (!g_capturing) { set_error("dev_tape_end: not capturing"); return QEMB_ERR_ARG; }
  g_capturing = false; ctx().tape_capture = false; ctx().in_region = false;
  std::vector<std::pair<hipGraphNode_t, int>> marks;
  marks.swap(ctx().marks);
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamEndCapture(g_stream, &graph));
  auto fail = [&](const std::string& why) { (void)hipGraphDestroy(graph); set_error("dev_tape_end: " + why); return 1; };     // 1: cannot tape (not an error)
  size_t nn = 0;
  if (hipGraphGetNodes(graph, nullptr, &nn) != hipSuccess) return fail("hipGraphGetNodes");
  std::vector<hipGraphNode_t> nodes(nn);
  if (nn && hipGraphGetNodes(graph, nodes.data(), &nn) != hipSuccess) return fail("hipGraphGetNodes");
  // execution order: a single-stream capture is a chain; follow the edges from the root
  size_t ne = 0;
  if (hipGraphGetEdges(graph, nullptr, nullptr, &ne) != hipSuccess) return fail("hipGraphGetEdges");
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne && hipGraphGetEdges(graph, from.data(), to.data(), &ne) != hipSuccess) return fail("hipGraphGetEdges");
  std::map<hipGraphNode_t, hipGraphNode_t> next;
  std::map<hipGraphNode_t, int> indeg;
  for (hipGraphNode_t x : nodes) indeg[x] = 0;
  for (size_t e = 0; e < ne; ++e) {
    if (next.count(from[e])) return fail("the captured sequence is not a chain");
    next[from[e]] = to[e]; indeg[to[e]] += 1;
  }
  hipGraphNode_t cur = nullptr;
  for (hipGraphNode_t x : nodes) if (indeg[x] == 0) { if (cur) return fail("the captured sequence has several roots"); cur = x; }
  Tape* t = new Tape();
  t->graph = graph;
  int cur_region = -1, cur_chain = -1, n_regions = 0;
  size_t mi = 0;
  auto apply_marks = [&](hipGraphNode_t last) {      // the marks recorded while `last` was the newest node of the capture
    for (; mi < marks.size() && marks[mi].first == last; ++mi) {
      const int v = marks[mi].second;
      if (v == MARK_REGION_BEGIN) { cur_region = n_regions++; cur_chain = -1; }
      else if (v == MARK_CHAIN) { if (cur_region >= 0) ++cur_chain; }
      else if (v == MARK_REGION_END) { cur_region = -1; cur_chain = -1; }
    }
  };
  apply_marks(nullptr);
  for (size_t visited = 0; cur && visited < nn; ++visited) {
    TapeNode tn{};
    if (hipGraphNodeGetType(cur, &tn.type) != hipSuccess) { delete t; return fail("hipGraphNodeGetType"); }
    if (tn.type == hipGraphNodeTypeKernel) {
      if (hipGraphKernelNodeGetParams(cur, &tn.k) != hipSuccess || !tn.k.func || !tn.k.kernelParams) { delete t; return fail("kernel node without parameters"); }
    } else if (tn.type == hipGraphNodeTypeMemcpy) {
      if (hipGraphMemcpyNodeGetParams(cur, &tn.cpy) != hipSuccess) { delete t; return fail("hipGraphMemcpyNodeGetParams"); }
    } else if (tn.type == hipGraphNodeTypeMemset) {
      if (hipGraphMemsetNodeGetParams(cur, &tn.set) != hipSuccess) { delete t; return fail("hipGraphMemsetNodeGetParams"); }
    } else if (tn.type != hipGraphNodeTypeEmpty) {
      delete t; return fail("node type " + std::to_string((int)tn.type) + " cannot be taped");
    }
    if (tn.type != hipGraphNodeTypeEmpty) { tn.region = cur_region; tn.chain = cur_region >= 0 ? cur_chain : -1; t->nodes.push_back(tn); }
    apply_marks(cur);
    auto it = next.find(cur);
    cur = (it == next.end()) ? nullptr : it->second;
  }
  if (mi != marks.size()) { delete t; return fail("a region mark does not lie on the captured chain"); }
  if (t->nodes.size() > nn) { delete t; return fail("node walk did not terminate"); }
  *out = t;
  return QEMB_OK;
}
struct TapePlan;
static void drop_plans_with(const void* tape);
int dev_tape_destroy(dev_tape_t tp) {
  Tape* t = (Tape*)tp;
  if (t) { drop_plans_with(t); if (t->graph) (void)hipGraphDestroy(t->graph); delete t; }
  return QEMB_OK;
}
int dev_tape_equal(dev_tape_t ap, dev_tape_t bp) {
  const Tape* a = (const Tape*)ap; const Tape* b = (const Tape*)bp;
  auto differ = [&](size_t i, const char* what) { set_error("tapes differ at operation " + std::to_string(i) + ": " + what); return 0; };
  if (!a || !b) return differ(0, "no tape");
  if (a->nodes.size() != b->nodes.size()) return differ(0, "number of operations");
  ensure_groupable_registered();
  std::vector<unsigned char> xa, xb;
  for (size_t i = 0; i < a->nodes.size(); ++i) {
    const TapeNode& x = a->nodes[i]; const TapeNode& y = b->nodes[i];
    if (x.type != y.type) return differ(i, "kind of operation");
    if (x.region != y.region || x.chain != y.chain) return differ(i, "region / chain");
    if (x.type == hipGraphNodeTypeKernel) {
      if (x.k.func != y.k.func) return differ(i, "kernel");
      if (x.k.gridDim.x != y.k.gridDim.x || x.k.gridDim.y != y.k.gridDim.y || x.k.gridDim.z != y.k.gridDim.z ||
          x.k.blockDim.x != y.k.blockDim.x || x.k.blockDim.y != y.k.blockDim.y || x.k.blockDim.z != y.k.blockDim.z || x.k.sharedMemBytes != y.k.sharedMemBytes) return differ(i, "launch geometry");
      auto it = groupable().find(x.k.func);
      if (it == groupable().end()) return differ(i, "a kernel whose argument list is not known to the grouped launches (not comparable)");
      const GroupInfo& gi = it->second;
      xa.assign(gi.args_bytes, 0); xb.assign(gi.args_bytes, 0);
      GroupMember ma{x.k.kernelParams, x.k.gridDim.x, x.k.gridDim.y, x.k.gridDim.z}, mb{y.k.kernelParams, y.k.gridDim.x, y.k.gridDim.y, y.k.gridDim.z};
      (void)gi.build(xa.data(), &ma, 1); (void)gi.build(xb.data(), &mb, 1);
      for (size_t q = 0; q < gi.args_bytes; ++q) if (xa[q] != xb[q]) return differ(i, ("kernel argument byte " + std::to_string(q) + " of " + std::to_string(gi.args_bytes)).c_str());
    } else if (x.type == hipGraphNodeTypeMemset) {
      if (x.set.dst != y.set.dst || x.set.value != y.set.value || x.set.width != y.set.width || x.set.height != y.set.height || x.set.elementSize != y.set.elementSize) return differ(i, "memset");
    } else if (x.type == hipGraphNodeTypeMemcpy) {
      if (std::memcmp(&x.cpy, &y.cpy, sizeof(x.cpy)) != 0) return differ(i, "copy");
    }
  }
  return 1;
}
static int tape_issue_single(const TapeNode& n, hipStream_t s) {
  if (n.type == hipGraphNodeTypeKernel) {
    HIP_TRY(hipLaunchKernel(n.k.func, n.k.gridDim, n.k.blockDim, n.k.kernelParams, n.k.sharedMemBytes, s));
  } else if (n.type == hipGraphNodeTypeMemcpy) {
    // (a captured 1-D hipMemcpyAsync comes back as 3-D parameters with extent {bytes, 1, 1}; hipMemcpy3DAsync refuses them -- invalid argument -- so the
    //  linear case is replayed as the linear copy it was)
    const hipMemcpy3DParms& c = n.cpy;
    if (c.extent.height <= 1 && c.extent.depth <= 1 && !c.srcArray && !c.dstArray && c.dstPtr.ptr && c.srcPtr.ptr && c.extent.width > 0)
      HIP_TRY(hipMemcpyAsync((char*)c.dstPtr.ptr + c.dstPos.x, (const char*)c.srcPtr.ptr + c.srcPos.x, c.extent.width, c.kind, s));
    else HIP_TRY(hipMemcpy3DAsync(&n.cpy, s));
  } else if (n.type == hipGraphNodeTypeMemset) {
    if (n.set.height <= 1) {
      if (n.set.elementSize == 1) HIP_TRY(hipMemsetAsync(n.set.dst, (int)n.set.value, n.set.width, s));
      else if (n.set.elementSize == 4) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)n.set.dst, (int)n.set.value, n.set.width, s));
      else { set_error("dev_tape_run: memset element size not handled"); return QEMB_ERR_ARG; }
    } else { set_error("dev_tape_run: 2-D memset not handled"); return QEMB_ERR_ARG; }
  }
  t_tape_launches += 1;
  return QEMB_OK;
}
// What one dev_tape_run of a given set of tapes issues: the launch plan is a function of the tapes alone, so it is built once per set
// (tables uploaded once) and replayed on every later run of the same set.
struct PlanStep {
  bool grouped;
  const TapeNode* single;       // !grouped
  const GroupInfo* gi; size_t args_off; unsigned blocks; dim3 block; size_t lds;     // grouped: its argument block inside TapePlan::args
};
struct TapePlan {
  std::vector<const void*> key;
  std::vector<PlanStep> steps;
  std::vector<unsigned char> args;      // argument blocks of the grouped launches (host memory: they are passed by value)
  long long ops = 0, grouped = 0;
};
static std::mutex g_plan_mutex;
static std::vector<TapePlan*> g_plans;        // small: one per distinct set of tapes that ran together

// one step of the plan from the nodes that may run together (same position of a plain segment, or one level of a parallel region): launches of the same
// kernel and block shape become grouped launches of up to GROUP_MAX members (arguments by value), the rest is issued singly.  (A member table in device
// memory with up to 64 members per launch was measured in round 5: six fragments x two chains in one launch were 0.5 ms per octane BE2 sweep SLOWER than two
// launches of six -- the table is chased through memory by every workgroup where the by-value arguments are scalar loads.)
static void plan_emit(TapePlan* plan, const std::vector<const TapeNode*>& here, bool grouping) {
  std::vector<unsigned char>& host = plan->args;
  auto reserve = [&](std::vector<unsigned char>& v, size_t bytes) { const size_t off = (v.size() + 255) / 256 * 256; v.resize(off + bytes); return off; };
  plan->ops += (long long)here.size();
  std::vector<bool> done(here.size(), false);
  for (size_t a = 0; a < here.size(); ++a) {
    if (done[a]) continue;
    const TapeNode* na = here[a];
    std::vector<const TapeNode*> grp{na};
    const GroupInfo* gi = nullptr;
    const int cap = GROUP_MAX;
    if (grouping && na->type == hipGraphNodeTypeKernel) {
      auto it = groupable().find(na->k.func);
      if (it != groupable().end()) {
        gi = &it->second;
        for (size_t b = a + 1; b < here.size() && (int)grp.size() < cap; ++b) {
          const TapeNode* nb = here[b];
          if (!done[b] && nb->type == hipGraphNodeTypeKernel && nb->k.func == na->k.func && nb->k.blockDim.x == na->k.blockDim.x &&
              nb->k.blockDim.y == na->k.blockDim.y && nb->k.blockDim.z == na->k.blockDim.z) { grp.push_back(nb); done[b] = true; }
        }
      }
    }
    done[a] = true;
    PlanStep st{};
    if (gi && grp.size() >= 2) {
      st.grouped = true; st.gi = gi; st.block = na->k.blockDim; st.lds = 0;
      std::vector<GroupMember> mem;
      for (const TapeNode* x : grp) {
        mem.push_back(GroupMember{x->k.kernelParams, x->k.gridDim.x, x->k.gridDim.y, x->k.gridDim.z});
        st.lds = std::max<size_t>(st.lds, x->k.sharedMemBytes);
      }
      st.args_off = reserve(host, gi->args_bytes);
      st.blocks = gi->build(host.data() + st.args_off, mem.data(), (int)mem.size());
      plan->grouped += 1;
      plan->steps.push_back(st);
    } else {
      for (const TapeNode* x : grp) { PlanStep s1{}; s1.grouped = false; s1.single = x; plan->steps.push_back(s1); }
    }
  }
}

// A tape as a sequence of segments: runs of plain nodes, and parallel regions (their chains).  Tapes of one run are merged segment by segment when their
// segment structure agrees (same code path for every fragment); otherwise position by position as if there were no regions.
struct TapeSegment { int region; size_t begin, end; std::vector<std::vector<const TapeNode*>> chains; };
static std::vector<TapeSegment> tape_segments(const Tape* t) {
  std::vector<TapeSegment> out;
  for (size_t i = 0; i < t->nodes.size(); ++i) {
    const TapeNode& nd = t->nodes[i];
    if (out.empty() || out.back().region != nd.region) out.push_back(TapeSegment{nd.region, i, i, {}});
    TapeSegment& sg = out.back();
    sg.end = i + 1;
    if (nd.region >= 0) {
      if ((int)sg.chains.size() <= nd.chain) sg.chains.resize((size_t)nd.chain + 1);
      sg.chains[(size_t)nd.chain].push_back(&nd);
    }
  }
  return out;
}

static int build_plan(const dev_tape_t* tapes, int n, TapePlan** out) {
  ensure_groupable_registered();
  static const bool grouping = !(std::getenv("QEMB_TAPE_GROUP") && std::atoi(std::getenv("QEMB_TAPE_GROUP")) == 0);
  const bool regions_on = regions_enabled();
  TapePlan* plan = new TapePlan();
  for (int f = 0; f < n; ++f) plan->key.push_back(tapes[f]);
  std::vector<std::vector<TapeSegment>> segs;
  bool same_structure = regions_on;
  for (int f = 0; f < n && same_structure; ++f) {
    segs.push_back(tape_segments((const Tape*)tapes[f]));
    if (f > 0) {
      same_structure = segs[f].size() == segs[0].size();
      for (size_t k = 0; same_structure && k < segs[0].size(); ++k)
        same_structure = (segs[f][k].region >= 0) == (segs[0][k].region >= 0) && segs[f][k].chains.size() == segs[0][k].chains.size();
    }
  }
  if (same_structure && !segs.empty()) {
    for (size_t k = 0; k < segs[0].size(); ++k) {
      if (segs[0][k].region < 0) {      // plain segment: position by position
        size_t longest = 0;
        for (int f = 0; f < n; ++f) longest = std::max(longest, segs[f][k].end - segs[f][k].begin);
        for (size_t i = 0; i < longest; ++i) {
          std::vector<const TapeNode*> here;
          for (int f = 0; f < n; ++f) { const Tape* t = (const Tape*)tapes[f]; const size_t j = segs[f][k].begin + i; if (j < segs[f][k].end) here.push_back(&t->nodes[j]); }
          plan_emit(plan, here, grouping);
        }
      } else {                          // parallel region: level by level over every chain of every tape
        size_t depth = 0;
        for (int f = 0; f < n; ++f) for (const auto& c : segs[f][k].chains) depth = std::max(depth, c.size());
        for (size_t lev = 0; lev < depth; ++lev) {
          std::vector<const TapeNode*> here;
          for (int f = 0; f < n; ++f) for (const auto& c : segs[f][k].chains) if (lev < c.size()) here.push_back(c[lev]);
          plan_emit(plan, here, grouping);
        }
      }
    }
  } else {
    size_t longest = 0;
    for (int f = 0; f < n; ++f) longest = std::max(longest, ((const Tape*)tapes[f])->nodes.size());
    for (size_t i = 0; i < longest; ++i) {
      std::vector<const TapeNode*> here;
      for (int f = 0; f < n; ++f) { const Tape* t = (const Tape*)tapes[f]; if (i < t->nodes.size()) here.push_back(&t->nodes[i]); }
      plan_emit(plan, here, grouping);
    }
  }
  *out = plan;
  return QEMB_OK;
}
int dev_tape_run(const dev_tape_t* tapes, int n) {
  REQUIRE_INIT();
  if (g_capturing) { set_error("dev_tape_run: inside a capture"); return QEMB_ERR_ARG; }
  if (n <= 0) return QEMB_OK;
  TapePlan* plan = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    for (TapePlan* p : g_plans) {
      if ((int)p->key.size() != n) continue;
      bool same = true;
      for (int f = 0; f < n && same; ++f) same = (p->key[f] == tapes[f]);
      if (same) { plan = p; break; }
    }
    if (!plan) {
      int rc = build_plan(tapes, n, &plan);
      if (rc) return rc;
      g_plans.push_back(plan);
    }
  }
  t_tape_launches = 0; t_tape_grouped = plan->grouped; t_tape_ops = plan->ops;
  for (const PlanStep& st : plan->steps) {
    if (st.grouped) {
      st.gi->launch(plan->args.data() + st.args_off, st.blocks, st.block, st.lds, g_stream);
      t_tape_launches += 1;
    } else {
      int rc = tape_issue_single(*st.single, g_stream);
      if (rc) return rc;
    }
  }
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
static void drop_plans_with(const void* tape) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  for (size_t i = 0; i < g_plans.size();) {
    TapePlan* p = g_plans[i];
    if (std::find(p->key.begin(), p->key.end(), tape) != p->key.end()) {
      delete p; g_plans.erase(g_plans.begin() + i);
    } else ++i;
  }
}
int dev_tape_last_stats(long long* launches, long long* grouped, long long* operations) {
  if (launches) *launches = t_tape_launches;
  if (grouped) *grouped = t_tape_grouped;
  if (operations) *operations = t_tape_ops;
  return QEMB_OK;
}

// ---- timers -----------------------------------------------------------------------------------
static bool timers_enabled() {
  static const bool on = [] { const char* e = std::getenv("QEMB_TIMERS"); return !(e && e[0] == '0'); }();
  return on;
}
// fold every lap whose stop event has completed into the totals (oldest first; stops at the first one still in flight unless `wait`)
static void timer_harvest(TimerSlot& t, bool wait) {
  size_t done = 0;
  for (; done < t.pending.size(); ++done) {
    TimerLap& lap = t.pending[done];
    if (lap.ended) {
      if (!wait && hipEventQuery(lap.e1) != hipSuccess) { (void)hipGetLastError(); break; }
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, lap.e0, lap.e1) == hipSuccess) { t.total_ms += ms; t.count += 1; }
      else (void)hipGetLastError();
    }                                   // a lap that was begun but never ended is dropped
    t.spare.emplace_back(lap.e0, lap.e1);
  }
  t.pending.erase(t.pending.begin(), t.pending.begin() + done);
}
int dev_timer_begin(int slot) {
  REQUIRE_INIT();
  if (g_capturing || !timers_enabled()) return QEMB_OK;
  if (slot < 0 || slot >= TIMER_NSLOTS) return QEMB_ERR_ARG;
  TimerSlot& t = g_timers[slot];
  if (!t.pending.empty() && !t.pending.back().ended) {     // the previous region left early: reuse its pair
    HIP_TRY(hipEventRecord(t.pending.back().e0, g_stream));
    return QEMB_OK;
  }
  if (t.pending.size() >= TIMER_MAX_PENDING) {
    timer_harvest(t, false);
    if (t.pending.size() >= TIMER_MAX_PENDING) { HIP_TRY(hipStreamSynchronize(g_stream)); timer_harvest(t, true); }
  }
  TimerLap lap{nullptr, nullptr, false};
  if (!t.spare.empty()) { lap.e0 = t.spare.back().first; lap.e1 = t.spare.back().second; t.spare.pop_back(); }
  else {
    HIP_TRY(hipEventCreate(&lap.e0));
    if (hipEventCreate(&lap.e1) != hipSuccess) { (void)hipEventDestroy(lap.e0); set_error("hipEventCreate failed"); return QEMB_ERR_DEVICE; }
  }
  t.pending.push_back(lap);
  HIP_TRY(hipEventRecord(lap.e0, g_stream));
  return QEMB_OK;
}
int dev_timer_end(int slot) {
  REQUIRE_INIT();
  if (g_capturing || !timers_enabled()) return QEMB_OK;
  if (slot < 0 || slot >= TIMER_NSLOTS) return QEMB_ERR_ARG;
  TimerSlot& t = g_timers[slot];
  if (t.pending.empty() || t.pending.back().ended) return QEMB_ERR_ARG;
  HIP_TRY(hipEventRecord(t.pending.back().e1, g_stream));
  t.pending.back().ended = true;
  return QEMB_OK;
}
static int timer_collect(DevCtx& c, int slot, double* total_ms, int64_t* count) {
  TimerSlot& t = c.timers[slot];
  HIP_TRY(hipStreamSynchronize(c.stream));
  timer_harvest(t, true);
  if (total_ms) *total_ms = t.total_ms;
  if (count) *count = t.count;
  return QEMB_OK;
}
int dev_timer_read(int slot, double* total_ms, int64_t* count) {
  REQUIRE_INIT();
  if (slot < 0 || slot >= TIMER_NSLOTS) return QEMB_ERR_ARG;
  return timer_collect(ctx(), slot, total_ms, count);
}
// timers of context k (call while no thread is driving that context); reset != 0 also clears them
int dev_ctx_timer_read(int k, int slot, double* total_ms, int64_t* count, int reset) {
  REQUIRE_INIT();
  if (slot < 0 || slot >= TIMER_NSLOTS) return QEMB_ERR_ARG;
  DevCtx* c = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (k < 0 || k > (int)g_extra_ctx.size()) { set_error("dev_ctx_timer_read: no such context"); return QEMB_ERR_ARG; }
    c = (k == 0) ? &g_default_ctx : g_extra_ctx[k - 1];
  }
  int rc = timer_collect(*c, slot, total_ms, count);
  if (rc == QEMB_OK && reset) { c->timers[slot].total_ms = 0; c->timers[slot].count = 0; }
  return rc;
}
int dev_timer_reset(int slot) {
  double a; int64_t c;
  int rc = dev_timer_read(slot, &a, &c);
  if (rc) return rc;
  g_timers[slot].total_ms = 0; g_timers[slot].count = 0;
  return QEMB_OK;
}
// number of live (pending + spare) event pairs of the calling context's slot: a test hook for the bounded-events guarantee
int dev_timer_live_events(int slot) {
  if (slot < 0 || slot >= TIMER_NSLOTS) return -1;
  return (int)(g_timers[slot].pending.size() + g_timers[slot].spare.size());
}

// ------------------------------------------------------------------------------------------------
// absolute overlap of primitive Cartesian Gaussians (screening matrix of the semi-sparse DF pipeline)
// ------------------------------------------------------------------------------------------------
// One workgroup per primitive shell pair (i >= j).  Per Cartesian direction d the quadrature sums I_d[p][q] = sum_n w_n |xa_d^p xb_d^q|
// (p <= l_i, q <= l_j) are accumulated by the threads over the roots and reduced across the workgroup; every component pair of the
// two shells is then the product Ix[ix][jx] Iy[iy][jy] Iz[iz][jz] times the Gaussian-product prefactor.
__global__ void __launch_bounds__(256) abs_overlap_prim_kernel(int nsh, const int* __restrict__ l, const double* __restrict__ ex,
                                                               const double* __restrict__ xyz, const long long* __restrict__ cart0,
                                                               long long ncart, int nroots, const double* __restrict__ roots,
                                                               const double* __restrict__ weights, double* __restrict__ out) {
  constexpr int LM = 5;   // l <= 4
  __shared__ double part[4][3 * LM * LM];
  __shared__ double I[3][LM][LM];
  const int i = blockIdx.x, j = blockIdx.y;
  if (j > i) return;
  const int li = l[i], lj = l[j];
  const double ai = ex[i], aj = ex[j], aij = ai + aj, scale = 1.0 / sqrt(aij);
  double Ra[3], Rb[3], Rp[3], r2 = 0.0;
  for (int d = 0; d < 3; ++d) {
    Ra[d] = xyz[3 * i + d]; Rb[d] = xyz[3 * j + d];
    Rp[d] = (ai * Ra[d] + aj * Rb[d]) / aij;
    r2 += (Ra[d] - Rb[d]) * (Ra[d] - Rb[d]);
  }
  double acc[3][LM][LM];
  for (int d = 0; d < 3; ++d) for (int p = 0; p < LM; ++p) for (int q = 0; q < LM; ++q) acc[d][p][q] = 0.0;
  for (int n = threadIdx.x; n < nroots; n += blockDim.x) {
    const double w = weights[n], t = roots[n] * scale;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double xa = fabs(t + Rp[d] - Ra[d]), xb = fabs(t + Rp[d] - Rb[d]);
      double pa = 1.0;
#pragma unroll
      for (int p = 0; p < LM; ++p) {
        double pb = pa * w;
#pragma unroll
        for (int q = 0; q < LM; ++q) { acc[d][p][q] += pb; pb *= xb; }
        pa *= xa;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = 0; d < 3; ++d) for (int p = 0; p < LM; ++p) for (int q = 0; q < LM; ++q) {
    double v = acc[d][p][q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) part[wave][(d * LM + p) * LM + q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3 * LM * LM) {
    const int t = threadIdx.x;
    (&I[0][0][0])[t] = part[0][t] + part[1][t] + part[2][t] + part[3][t];
  }
  __syncthreads();
  const double fac = scale * scale * scale * exp(-(ai * aj / aij) * r2);
  const int nfi = (li + 1) * (li + 2) / 2, nfj = (lj + 1) * (lj + 2) / 2;
  for (int t = threadIdx.x; t < nfi * nfj; t += blockDim.x) {
    const int ci = t / nfj, cj = t % nfj;
    // component index -> (lx, ly, lz) in libcint order: lx from l down to 0, ly from l - lx down to 0
    int ix = li, iy = 0, iz = 0, c = ci;
    for (ix = li; ix >= 0; --ix) { const int cnt = li - ix + 1; if (c < cnt) { iy = li - ix - c; iz = li - ix - iy; break; } c -= cnt; }
    int jx = lj, jy = 0, jz = 0; c = cj;
    for (jx = lj; jx >= 0; --jx) { const int cnt = lj - jx + 1; if (c < cnt) { jy = lj - jx - c; jz = lj - jx - jy; break; } c -= cnt; }
    const double v = I[0][ix][jx] * I[1][iy][jy] * I[2][iz][jz] * fac;
    out[(cart0[i] + ci) * ncart + cart0[j] + cj] = v;
    out[(cart0[j] + cj) * ncart + cart0[i] + ci] = v;
  }
}
int dev_abs_overlap_prim(int nsh, const int* l, const double* ex, const double* xyz, const int64_t* cart0, int64_t ncart, int nroots,
                         const double* roots, const double* weights, double* out) {
  REQUIRE_INIT();
  if (nsh <= 0) return QEMB_OK;
  hipLaunchKernelGGL(abs_overlap_prim_kernel, dim3((unsigned)nsh, (unsigned)nsh), dim3(256), 0, g_stream, nsh, l, ex, xyz,
                     (const long long*)cart0, (long long)ncart, nroots, roots, weights, out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// fill
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fill_kernel_body(const uint3 BID, const uint3 GDIM, double* x, long long n, double v) {
  long long i = (long long)BID.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)GDIM.x * blockDim.x;
  for (; i < n; i += stride) x[i] = v;
}
__global__ void fill_kernel(double* x, long long n, double v) { fill_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), x, n, v); }
int dev_fill(double* x, int64_t n, double value) {
  REQUIRE_INIT();
  if (n <= 0) return QEMB_OK;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, g_stream, x, (long long)n, value);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// strided 4-d copy/add.  Two kernels: "linear" (threads run along the dimension that is contiguous on
// the input side -- also contiguous on the output side when no transpose is involved) and "transpose"
// (32x32 LDS tile over the input-contiguous x output-contiguous pair of dimensions).
// ------------------------------------------------------------------------------------------------
struct Copy4K {
  long long d0, d1, d2, d3;
  long long si0, si1, si2, si3, so0, so1, so2, so3;
  const double* in; double* out; double alpha, beta; const double* base;
  double* out2; const double* in2; double c2a, c2b;      // optional second output of the pass (see Copy4Desc)
};

__device__ __forceinline__ void copy4_linear_kernel_body(const uint3 BID, const uint3 GDIM, Copy4K c) {
  // x: combined (i2,i3) index, y: i1, z: i0 (both looped)
  const long long n23 = c.d2 * c.d3;
  for (long long i0 = BID.z; i0 < c.d0; i0 += GDIM.z) {
    for (long long i1 = BID.y; i1 < c.d1; i1 += GDIM.y) {
      const long long bi = i0 * c.si0 + i1 * c.si1, bo = i0 * c.so0 + i1 * c.so1;
      for (long long t = (long long)BID.x * blockDim.x + threadIdx.x; t < n23;
           t += (long long)GDIM.x * blockDim.x) {
        const long long i2 = t / c.d3, i3 = t - i2 * c.d3;
        const double v = c.alpha * c.in[bi + i2 * c.si2 + i3 * c.si3];
        const long long off = bo + i2 * c.so2 + i3 * c.so3;
        const double w = (c.beta != 0.0) ? v + c.beta * c.base[off] : v;
        c.out[off] = w;
        if (c.out2) c.out2[off] = c.c2a * c.in2[off] + c.c2b * w;
      }
    }
  }
}
__global__ void __launch_bounds__(256) copy4_linear_kernel(Copy4K c) { copy4_linear_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), c); }

// dims canonicalised so that si3 == 1 (input contiguous along i3) and so2 == 1 (output contiguous along i2)
__device__ __forceinline__ void copy4_transpose_kernel_body(const uint3 BID, const uint3 GDIM, Copy4K c, int tiles3) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const uint3 LB = xcd_logical_block(BID, GDIM);      // neighbouring tiles of a slab on one XCD (their row pieces share 128-byte lines)
  const long long t2 = LB.x / tiles3, t3 = LB.x % tiles3;
  const long long base2 = t2 * 32, base3 = t3 * 32;
  for (long long i0 = LB.z; i0 < c.d0; i0 += GDIM.z) {
    for (long long i1 = LB.y; i1 < c.d1; i1 += GDIM.y) {
      const long long bi = i0 * c.si0 + i1 * c.si1, bo = i0 * c.so0 + i1 * c.so1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long i2 = base2 + ty + 8 * r, i3 = base3 + tx;
        if (i2 < c.d2 && i3 < c.d3) tile[ty + 8 * r][tx] = c.in[bi + i2 * c.si2 + i3 * c.si3];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long i3 = base3 + ty + 8 * r, i2 = base2 + tx;
        if (i2 < c.d2 && i3 < c.d3) {
          const double v = c.alpha * tile[tx][ty + 8 * r];
          const long long off = bo + i2 * c.so2 + i3 * c.so3;
          const double w = (c.beta != 0.0) ? v + c.beta * c.base[off] : v;
          c.out[off] = w;
          if (c.out2) c.out2[off] = c.c2a * c.in2[off] + c.c2b * w;
        }
      }
      __syncthreads();
    }
  }
}
__global__ void __launch_bounds__(256) copy4_transpose_kernel(Copy4K c, int tiles3) { copy4_transpose_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), c, tiles3); }

int dev_copy4(const Copy4Desc& cd) {
  REQUIRE_INIT();
  for (int k = 0; k < 4; ++k) if (cd.dim[k] <= 0) return QEMB_OK;
  if (!cd.in || !cd.out) { set_error("dev_copy4: null pointer"); return QEMB_ERR_ARG; }
  // canonicalise the loop order: input-contiguous dim -> slot 3, output-contiguous dim -> slot 2
  int order[4] = {0, 1, 2, 3};
  int a = -1, b = -1;
  for (int k = 3; k >= 0; --k) if (cd.si[k] == 1 && cd.dim[k] > 1) { a = k; break; }
  for (int k = 3; k >= 0; --k) if (cd.so[k] == 1 && cd.dim[k] > 1) { b = k; break; }
  bool transpose = (a >= 0 && b >= 0 && a != b && cd.dim[a] >= 8 && cd.dim[b] >= 8);
  if (a < 0) a = 3;
  {
    std::vector<int> rest;
    if (transpose) { for (int k = 0; k < 4; ++k) if (k != a && k != b) rest.push_back(k); order[0] = rest[0]; order[1] = rest[1]; order[2] = b; order[3] = a; }
    else { for (int k = 0; k < 4; ++k) if (k != a) rest.push_back(k); order[0] = rest[0]; order[1] = rest[1]; order[2] = rest[2]; order[3] = a; }
  }
  Copy4K c{};
  long long d[4], si[4], so[4];
  for (int k = 0; k < 4; ++k) { d[k] = cd.dim[order[k]]; si[k] = cd.si[order[k]]; so[k] = cd.so[order[k]]; }
  c.d0 = d[0]; c.d1 = d[1]; c.d2 = d[2]; c.d3 = d[3];
  c.si0 = si[0]; c.si1 = si[1]; c.si2 = si[2]; c.si3 = si[3];
  c.so0 = so[0]; c.so1 = so[1]; c.so2 = so[2]; c.so3 = so[3];
  c.in = cd.in; c.out = cd.out; c.alpha = cd.alpha; c.beta = cd.beta; c.base = cd.base ? cd.base : cd.out;
  c.out2 = cd.out2; c.in2 = cd.in2; c.c2a = cd.c2a; c.c2b = cd.c2b;
  if (c.out2 && !c.in2) { set_error("dev_copy4: second output without its input"); return QEMB_ERR_ARG; }
  if (c.d2 * c.d3 >= (1LL << 40)) { set_error("dev_copy4: inner extent too large"); return QEMB_ERR_ARG; }
  const unsigned gy = (unsigned)std::min<long long>(c.d1, 65535), gz = (unsigned)std::min<long long>(c.d0, 65535);
  if (transpose) {
    const long long tiles2 = (c.d2 + 31) / 32, tiles3 = (c.d3 + 31) / 32;
    if (tiles2 * tiles3 > 0x7fffffffLL) { set_error("dev_copy4: too many tiles"); return QEMB_ERR_ARG; }
    hipLaunchKernelGGL(copy4_transpose_kernel, dim3((unsigned)(tiles2 * tiles3), gy, gz), dim3(256), 0, g_stream, c, (int)tiles3);
  } else {
    const long long n23 = c.d2 * c.d3;
    const unsigned gx = (unsigned)std::min<long long>((n23 + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(copy4_linear_kernel, dim3(gx, gy, gz), dim3(256), 0, g_stream, c);
  }
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// outer4 / denominators
// ------------------------------------------------------------------------------------------------
struct Outer4K {
  long long d0, d1, d2, d3, su0, su2, sv1, sv3, so0, so1, so2, so3;
  const double* u; const double* v; double* out; double alpha, beta; const double* base;
};
__device__ __forceinline__ void outer4_kernel_body(const uint3 BID, const uint3 GDIM, Outer4K c) {
  const long long n23 = c.d2 * c.d3;
  for (long long i0 = BID.z; i0 < c.d0; i0 += GDIM.z)
    for (long long i1 = BID.y; i1 < c.d1; i1 += GDIM.y)
      for (long long t = (long long)BID.x * blockDim.x + threadIdx.x; t < n23; t += (long long)GDIM.x * blockDim.x) {
        const long long i2 = t / c.d3, i3 = t - i2 * c.d3;
        const double val = c.alpha * c.u[i0 * c.su0 + i2 * c.su2] * c.v[i1 * c.sv1 + i3 * c.sv3];
        const long long off = i0 * c.so0 + i1 * c.so1 + i2 * c.so2 + i3 * c.so3;
        c.out[off] = (c.beta != 0.0) ? val + c.beta * c.base[off] : val;
      }
}
__global__ void __launch_bounds__(256) outer4_kernel(Outer4K c) { outer4_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), c); }
int dev_outer4(const Outer4Desc& o) {
  REQUIRE_INIT();
  for (int k = 0; k < 4; ++k) if (o.dim[k] <= 0) return QEMB_OK;
  Outer4K c{o.dim[0], o.dim[1], o.dim[2], o.dim[3], o.su0, o.su2, o.sv1, o.sv3, o.so[0], o.so[1], o.so[2], o.so[3], o.u, o.v, o.out, o.alpha, o.beta, o.base ? o.base : o.out};
  const long long n23 = c.d2 * c.d3;
  const unsigned gx = (unsigned)std::min<long long>((n23 + 255) / 256, 1 << 20);
  hipLaunchKernelGGL(outer4_kernel, dim3(gx, (unsigned)std::min<long long>(c.d1, 65535), (unsigned)std::min<long long>(c.d0, 65535)), dim3(256), 0, g_stream, c);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

__device__ __forceinline__ void div_denom_kernel_body(const uint3 BID, const uint3 GDIM, double* x, long long d0, long long d1, long long d2, long long d3,
                                                        const double* ea, const double* eb, const double* ec, const double* ed) {
  const long long n23 = d2 * d3;
  for (long long i0 = BID.z; i0 < d0; i0 += GDIM.z)
    for (long long i1 = BID.y; i1 < d1; i1 += GDIM.y) {
      const double e01 = ea[i0] + (eb ? eb[i1] : 0.0);
      double* row = x + (i0 * d1 + i1) * n23;
      for (long long t = (long long)BID.x * blockDim.x + threadIdx.x; t < n23; t += (long long)GDIM.x * blockDim.x) {
        const long long i2 = t / d3, i3 = t - i2 * d3;
        row[t] = row[t] / (e01 - ec[i2] - (ed ? ed[i3] : 0.0));
      }
    }
}
__global__ void __launch_bounds__(256) div_denom_kernel(double* x, long long d0, long long d1, long long d2, long long d3,
                                                        const double* ea, const double* eb, const double* ec, const double* ed) { div_denom_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), x, d0, d1, d2, d3, ea, eb, ec, ed); }
int dev_div_denom(double* x, int64_t d0, int64_t d1, int64_t d2, int64_t d3, const double* ea, const double* eb, const double* ec, const double* ed) {
  REQUIRE_INIT();
  if (d0 <= 0 || d1 <= 0 || d2 <= 0 || d3 <= 0) return QEMB_OK;
  const long long n23 = d2 * d3;
  const unsigned gx = (unsigned)std::min<long long>((n23 + 255) / 256, 1 << 20);
  hipLaunchKernelGGL(div_denom_kernel, dim3(gx, (unsigned)std::min<long long>(d1, 65535), (unsigned)std::min<long long>(d0, 65535)), dim3(256), 0, g_stream,
                     x, (long long)d0, (long long)d1, (long long)d2, (long long)d3, ea, eb, ec, ed);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ---- screening helpers (semi-sparse DF) --------------------------------------------------------------------------
__global__ void threshold_mask_kernel(long long n, const double* __restrict__ x, double eps, double* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (fabs(x[i]) >= eps) ? 1.0 : 0.0;
}
int dev_threshold_mask(int64_t n, const double* x, double eps, double* out) {
  REQUIRE_INIT();
  if (n <= 0) return QEMB_OK;
  hipLaunchKernelGGL(threshold_mask_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, g_stream, (long long)n, x, eps, out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) mul_bcast_rows_kernel(long long rows, long long cols, double* __restrict__ x, const double* __restrict__ m) {
  for (long long r = blockIdx.y; r < rows; r += gridDim.y)
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (long long)gridDim.x * blockDim.x)
      x[r * cols + c] *= m[c];
}
int dev_mul_bcast_rows(int64_t rows, int64_t cols, double* x, const double* m) {
  REQUIRE_INIT();
  if (rows <= 0 || cols <= 0) return QEMB_OK;
  hipLaunchKernelGGL(mul_bcast_rows_kernel, dim3((unsigned)std::min<int64_t>((cols + 255) / 256, 1024), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0, g_stream,
                     (long long)rows, (long long)cols, x, m);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// one 32 x 32 tile (c-range, b-range) of one (k, j) pair per workgroup: X = t2[k,j] is read as the tile and as the mirrored tile
// (for the transposed outputs, through LDS), every output tile is written in rows of 32 contiguous doubles
__device__ __forceinline__ void ccsd_ph_layouts_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, const double* __restrict__ t2, const double* __restrict__ t1,
                                                              double* __restrict__ T, double* __restrict__ Tp, double* __restrict__ S,
                                                              double* __restrict__ Ut, double* __restrict__ Tpt, double* __restrict__ Th, int tiles) {
  __shared__ double xt[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const uint3 LB = xcd_logical_block(BID, GDIM);
  const long long tc = LB.x / tiles, tb = LB.x % tiles;
  const long long k = LB.z, j = LB.y;
  const double* __restrict__ X = t2 + (k * o + j) * v * v;
  // The b-ranges of the tiles are shifted so that the 32-double runs written to the five [k,c,j,b] outputs start on 128-byte lines: their
  // rows start at ((k v + c) o + j) v doubles, which (when o v is a multiple of 16) is the same offset a = ((k v o + j) v) mod 16 into a line
  // for every c of the tile.  Runs that straddle lines are completed by another workgroup later and cost a read-modify-write at the memory
  // side: v = 200 ran at 3.8 TB/s against 4.9 for v = 192 / 208 before the shift (tools/align_experiment.py).
  const long long a = ((o * v) % 16 == 0) ? ((k * v * o + j) * v) % 16 : 0;
  const long long b0 = tb * 32 - a;
  // mirrored tile: rows b-range, columns c-range -> xt[b_local][c_local]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long b = b0 + ty + 8 * r, c = tc * 32 + tx;
    xt[ty + 8 * r][tx] = (b >= 0 && b < v && c < v) ? X[b * v + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long c = tc * 32 + ty + 8 * r, b = b0 + tx;
    if (c < v && b >= 0 && b < v) {
      const double x = X[c * v + b];                 // t2[k,j,c,b]
      const double xp = xt[tx][ty + 8 * r];          // t2[k,j,b,c]
      const double tt = 2.0 * t1[j * v + c] * t1[k * v + b];
      const long long off = ((k * v + c) * o + j) * v + b;
      T[off] = x;
      Tp[off] = xp;
      const double s = 2.0 * x - xp;
      S[off] = s;
      Ut[off] = s - tt;
      Tpt[off] = xp + tt;
      Th[((k * o + j) * v + c) * v + b] = 2.0 * xp - x;
    }
  }
}
__global__ void __launch_bounds__(256) ccsd_ph_layouts_kernel(long long o, long long v, const double* __restrict__ t2, const double* __restrict__ t1,
                                                              double* __restrict__ T, double* __restrict__ Tp, double* __restrict__ S,
                                                              double* __restrict__ Ut, double* __restrict__ Tpt, double* __restrict__ Th, int tiles) { ccsd_ph_layouts_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, t2, t1, T, Tp, S, Ut, Tpt, Th, tiles); }
int dev_ccsd_ph_layouts(int64_t o, int64_t v, const double* t2, const double* t1, double* T, double* Tp, double* S, double* Ut, double* Tpt, double* Th) {
  REQUIRE_INIT();
  if (o <= 0 || v <= 0) return QEMB_OK;
  if (o > 65535) { set_error("dev_ccsd_ph_layouts: too many occupied orbitals"); return QEMB_ERR_ARG; }
  const long long tiles_c = (v + 31) / 32, tiles = (v + 15 + 31) / 32;      // b-tiles: one more may be needed for the line-aligning shift (<= 15)
  hipLaunchKernelGGL(ccsd_ph_layouts_kernel, dim3((unsigned)(tiles_c * tiles), (unsigned)o, (unsigned)o), dim3(256), 0, g_stream, (long long)o, (long long)v, t2, t1,
                     T, Tp, S, Ut, Tpt, Th, (int)tiles);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

__device__ __forceinline__ void small_k_update_kernel_body(const uint3 BID, const uint3 GDIM, long long M, long long N, long long K, double alpha, const double* __restrict__ A, long long sA,
                                                             const double* __restrict__ B, long long sB, double* __restrict__ C, long long sC, int tiles_n) {
  __shared__ double As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long tm = BID.x / tiles_n, tn = BID.x % tiles_n;
  const long long m0 = tm * 32, n0 = tn * 32;
  {
    const long long z = BID.y;
    const double* __restrict__ Az = A + z * sA;
    const double* __restrict__ Bz = B + z * sB;
    double* __restrict__ Cz = C + z * sC;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long k = k0 + ty + 8 * r;
        As[ty + 8 * r][tx] = (k < K && m0 + tx < M) ? Az[k * M + m0 + tx] : 0.0;
        Bs[ty + 8 * r][tx] = (k < K && n0 + tx < N) ? Bz[k * N + n0 + tx] : 0.0;
      }
      __syncthreads();
      const int kc = (int)((K - k0 < 32) ? K - k0 : 32);
      for (int k = 0; k < kc; ++k) {
        const double b = Bs[k][tx];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += As[k][ty + 8 * r] * b;
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long m = m0 + ty + 8 * r, n = n0 + tx;
      if (m < M && n < N) Cz[m * N + n] += alpha * acc[r];
    }
  }
}
__global__ void __launch_bounds__(256) small_k_update_kernel(long long M, long long N, long long K, double alpha, const double* __restrict__ A, long long sA,
                                                             const double* __restrict__ B, long long sB, double* __restrict__ C, long long sC, int tiles_n) { small_k_update_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), M, N, K, alpha, A, sA, B, sB, C, sC, tiles_n); }
// The same update on the matrix pipe (M, N >= 16, K <= 64): 2 K M N flop per batch entry is
// 6.4 GFLOP for the rank-n_occ updates of the o^2 v^2 tensors -- 100 us of FP64 VALU time, which is what the tile version above and a
// VALU strip version both take, twice the HBM time of the 256 MB they move.  B[z] (K x N) and the
// A columns of the strip are small and cache resident; wave w owns the 16-column tiles w, w + 4, ... for both 16-row halves of the strip
// (v_mfma_f64_16x16x4_f64: A lane l holds A[row = l & 15][k = l >> 4], B lane l holds B[k = l >> 4][col = l & 15], D reg r of lane l is
// D[row = (l >> 4) + 4 r][col = l & 15]); the read-modify-write of C is 16 lanes x 8 B = 128 contiguous bytes per row.
typedef double d4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void small_k_update_mfma_kernel_body(const uint3 BID, const uint3 GDIM, int M, int N, int K, double alpha, const double* __restrict__ A, long long sA,
                                                                  const double* __restrict__ B, long long sB, double* __restrict__ C, long long sC) {
  // no LDS, no barrier: one wave per (32-row strip, 16-column tile); it fetches its fragments straight from memory (A and B are small and
  // cache resident) and has the C values it will update in flight while the MFMAs run
  const int NT = (N + 15) >> 4, MT = (M + 31) >> 5;
  const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
  const uint3 LB = xcd_logical_block(BID, GDIM);      // neighbouring column tiles of a strip (they share the lines their 128-byte row pieces straddle) on one XCD
  const int item = LB.x * 4 + (threadIdx.x >> 6);
  if (item >= MT * NT) return;
  const int m0 = (item / NT) * 32, col = (item % NT) * 16 + fr;
  const long long z = LB.y;
  const double* __restrict__ Az = A + z * sA;
  const double* __restrict__ Bz = B + z * sB;
  double* __restrict__ Cz = C + z * sC;
  const int ra = m0 + fr, rb = ra + 16;
  const bool cin = col < N;
  double c0[4], c1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ma = m0 + fk + 4 * r, mb = ma + 16;
    c0[r] = (cin && ma < M) ? Cz[(long long)ma * N + col] : 0.0;
    c1[r] = (cin && mb < M) ? Cz[(long long)mb * N + col] : 0.0;
  }
  d4v acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + fk;
    const bool kin = k < K;
    const double b = (kin && cin) ? Bz[(long long)k * N + col] : 0.0;
    const double a0 = (kin && ra < M) ? Az[(long long)k * M + ra] : 0.0;
    const double a1 = (kin && rb < M) ? Az[(long long)k * M + rb] : 0.0;
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
  }
  if (cin) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ma = m0 + fk + 4 * r, mb = ma + 16;
      if (ma < M) Cz[(long long)ma * N + col] = c0[r] + alpha * acc0[r];
      if (mb < M) Cz[(long long)mb * N + col] = c1[r] + alpha * acc1[r];
    }
  }
}
__global__ void __launch_bounds__(256) small_k_update_mfma_kernel(int M, int N, int K, double alpha, const double* __restrict__ A, long long sA,
                                                                  const double* __restrict__ B, long long sB, double* __restrict__ C, long long sC) { small_k_update_mfma_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), M, N, K, alpha, A, sA, B, sB, C, sC); }
int dev_small_k_update(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sA, const double* B, int64_t sB,
                       double* C, int64_t sC) {
  REQUIRE_INIT();
  if (batch <= 0 || M <= 0 || N <= 0 || K <= 0) return QEMB_OK;
  if (batch > 65535) { set_error("dev_small_k_update: batch too large"); return QEMB_ERR_ARG; }
  if (N >= 16 && M >= 16 && K <= 64 && M <= (1 << 20) && N <= (1 << 20)) {
    const long long items = ((M + 31) / 32) * ((N + 15) / 16);
    hipLaunchKernelGGL(small_k_update_mfma_kernel, dim3((unsigned)((items + 3) / 4), (unsigned)batch), dim3(256), 0, g_stream, (int)M, (int)N, (int)K, alpha,
                       A, (long long)sA, B, (long long)sB, C, (long long)sC);
    HIP_TRY(hipGetLastError());
    return QEMB_OK;
  }
  const long long tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32;
  hipLaunchKernelGGL(small_k_update_kernel, dim3((unsigned)(tiles_m * tiles_n), (unsigned)batch), dim3(256), 0, g_stream, (long long)M, (long long)N, (long long)K, alpha,
                     A, (long long)sA, B, (long long)sB, C, (long long)sC, (int)tiles_n);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// sum of the S split-K slabs of one element, in slab order (what the reduction pass forms); the common counts are spelled out so that their loads are
// independent instructions the scheduler can issue together (a counted loop made the ladder scatter wait for each slab in turn: 56 -> 172 us)
__device__ __forceinline__ double slab_sum(const double* __restrict__ p, int S, long long stride) {
  switch (S) {
    case 1: return p[0];
    case 2: { const double a = p[0], b = p[stride]; return a + b; }
    case 3: { const double a = p[0], b = p[stride], c = p[2 * stride]; return (a + b) + c; }
    case 4: { const double a = p[0], b = p[stride], c = p[2 * stride], d = p[3 * stride]; return ((a + b) + c) + d; }
    default: {
      double acc = 0.0;
      int sl = 0;
      for (; sl + 4 <= S; sl += 4) { const double a = p[sl * stride], b = p[(sl + 1) * stride], c = p[(sl + 2) * stride], d = p[(sl + 3) * stride]; acc = (((acc + a) + b) + c) + d; }
      for (; sl < S; ++sl) acc += p[sl * stride];
      return acc;
    }
  }
}
__device__ __forceinline__ void ccsd_y_traces_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, const double* __restrict__ ZC, const double* __restrict__ ZB,
                                                            double* __restrict__ Y, const double* __restrict__ add, int S, long long stride, double scale) {
  // eight lanes per output (a, c), lane q takes k = q, q + 8, ...: the ZB addresses of one output are v^2 o doubles apart (every load its own line), so the
  // pass is a matter of loads in flight -- a thread per output kept 3-4 workgroups per fragment busy walking n_occ dependent rounds
  const long long idx = ((long long)BID.x * blockDim.x + threadIdx.x) >> 3;
  const int q = threadIdx.x & 7;
  const bool in = idx < v * v;
  const long long a = in ? idx / v : 0, c = in ? idx % v : 0;
  double s = 0.0;
  if (in) {
    double zc[4], zb[4];
    long long k = q;
    for (; k + 24 < o; k += 32) {      // four terms' loads in flight
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long long kk = k + 8 * u; zc[u] = ZC[((kk * o + kk) * v + a) * v + c]; zb[u] = ZB[((kk * v + c) * v + a) * o + kk]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) s += 2.0 * zc[u] - zb[u];
    }
    for (; k < o; k += 8) s += 2.0 * ZC[((k * o + k) * v + a) * v + c] - ZB[((k * v + c) * v + a) * o + k];
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (in && q == 0) Y[idx] = add ? s + scale * slab_sum(add + idx, S, stride) : s;
}
__global__ void __launch_bounds__(256) ccsd_y_traces_kernel(long long o, long long v, const double* __restrict__ ZC, const double* __restrict__ ZB,
                                                            double* __restrict__ Y, const double* __restrict__ add, int S, long long stride, double scale) { ccsd_y_traces_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, ZC, ZB, Y, add, S, stride, scale); }
int dev_ccsd_y_traces(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add, int S, int64_t stride, double scale) {
  REQUIRE_INIT();
  if (v <= 0) return QEMB_OK;
  hipLaunchKernelGGL(ccsd_y_traces_kernel, dim3((unsigned)((v * v * 8 + 255) / 256)), dim3(256), 0, g_stream, (long long)o, (long long)v, ZC, ZB, Y, add, std::max(S, 1), (long long)stride, scale);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// one row per blockIdx.y (grid-stride), 256 threads stride along the row: reads and writes are contiguous runs of `len` doubles
__global__ void __launch_bounds__(256) gather_rows_kernel(long long nrows, long long len, const long long* __restrict__ idx,
                                                          const double* __restrict__ src, long long ld, double* __restrict__ dst) {
  for (long long r = blockIdx.y; r < nrows; r += gridDim.y) {
    const long long sr = idx[r];
    const double* in = src + (sr < 0 ? 0 : sr) * ld;
    double* out = dst + r * len;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < len; c += (long long)gridDim.x * blockDim.x)
      out[c] = sr < 0 ? 0.0 : in[c];
  }
}
int dev_gather_rows(int64_t nrows, int64_t len, const int64_t* idx_dev, const double* src, int64_t ld, double* dst) {
  REQUIRE_INIT();
  if (nrows <= 0 || len <= 0) return QEMB_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 64), (unsigned)std::min<int64_t>(nrows, 65535)), dim3(256), 0, g_stream,
                     (long long)nrows, (long long)len, (const long long*)idx_dev, src, (long long)ld, dst);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) scale_rows_kernel(long long nrows, long long len, double* __restrict__ x, const double* __restrict__ s) {
  for (long long r = blockIdx.y; r < nrows; r += gridDim.y) {
    const double f = s[r];
    if (f == 1.0) continue;
    double* row = x + r * len;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < len; c += (long long)gridDim.x * blockDim.x) row[c] = (f == 0.0) ? 0.0 : row[c] * f;
  }
}
int dev_scale_rows(int64_t nrows, int64_t len, double* x, const double* s) {
  REQUIRE_INIT();
  if (nrows <= 0 || len <= 0) return QEMB_OK;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 64), (unsigned)std::min<int64_t>(nrows, 65535)), dim3(256), 0, g_stream,
                     (long long)nrows, (long long)len, x, s);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

__global__ void __launch_bounds__(256) mirror_lower_kernel(long long n, double* __restrict__ A, long long lda) {
  __shared__ double tile[32][33];
  // the logical block number enumerates lower-triangle 32 x 32 tiles (tr >= tc), neighbours of a tile row on one XCD (xcd_logical_block)
  long long t = xcd_logical_block(make_uint3(blockIdx.x, 0, 0), make_uint3(gridDim.x, 1, 1)).x;
  long long tr = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (tr * (tr + 1) / 2 > t) --tr;
  while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
  const long long tc = t - tr * (tr + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long r = tr * 32 + ty + 8 * i, c = tc * 32 + tx;
    tile[ty + 8 * i][tx] = (r < n && c < n) ? A[r * lda + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long r = tc * 32 + ty + 8 * i, c = tr * 32 + tx;   // destination (upper) element (r, c) = source (c, r)
    if (r < n && c < n && r < c) A[r * lda + c] = tile[tx][ty + 8 * i];
  }
}
int dev_mirror_lower(int64_t n, double* A, int64_t lda) {
  REQUIRE_INIT();
  if (n <= 1) return QEMB_OK;
  const long long nt = (n + 31) / 32;
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, g_stream, (long long)n, A, (long long)lda);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

struct LincombK { const double* x[8]; double c[8]; int n; };
__device__ __forceinline__ void lincomb_kernel_body(const uint3 BID, const uint3 GDIM, long long n, LincombK k, double beta, double* __restrict__ out) {
  for (long long t = (long long)BID.x * blockDim.x + threadIdx.x; t < n; t += (long long)GDIM.x * blockDim.x) {
    double acc = (beta != 0.0) ? beta * out[t] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q < k.n) acc += k.c[q] * k.x[q][t];
    out[t] = acc;
  }
}
__global__ void __launch_bounds__(256) lincomb_kernel(long long n, LincombK k, double beta, double* __restrict__ out) { lincomb_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), n, k, beta, out); }
int dev_lincomb(int64_t n, int nterms, const double* coef, const double* const* xs, double beta, double* out) {
  REQUIRE_INIT();
  if (n <= 0) return QEMB_OK;
  if (nterms < 0 || nterms > 8) { set_error("dev_lincomb: at most 8 terms"); return QEMB_ERR_ARG; }
  LincombK k{};
  k.n = nterms;
  for (int q = 0; q < nterms; ++q) { k.x[q] = xs[q]; k.c[q] = coef[q]; }
  const long long blocks = std::min<long long>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)blocks), dim3(256), 0, g_stream, (long long)n, k, beta, out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ---- pair-packed MO transformation helpers --------------------------------------------------------------------------
__device__ __forceinline__ long long pair_idx(long long i, long long j);
__device__ __forceinline__ void unpair_ge(long long p, long long& x, long long& y) {   // p = x(x+1)/2 + y, x >= y
  x = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
  while (x * (x + 1) / 2 > p) --x;
  while ((x + 1) * (x + 2) / 2 <= p) ++x;
  y = p - x * (x + 1) / 2;
}
__global__ void __launch_bounds__(256) pack_pair_rows_kernel(long long n, long long ncols, const double* __restrict__ in, double* __restrict__ out) {
  const long long np = n * (n + 1) / 2;
  for (long long p = blockIdx.y; p < np; p += gridDim.y) {
    long long x, y; unpair_ge(p, x, y);
    const double* src = in + (x * n + y) * ncols;
    double* dst = out + p * ncols;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (long long)gridDim.x * blockDim.x) dst[c] = src[c];
  }
}
int dev_pack_pair_rows(int64_t n, int64_t ncols, const double* in, double* out) {
  REQUIRE_INIT();
  const long long np = n * (n + 1) / 2;
  if (np <= 0 || ncols <= 0) return QEMB_OK;
  hipLaunchKernelGGL(pack_pair_rows_kernel, dim3((unsigned)std::min<long long>((ncols + 255) / 256, 256), (unsigned)std::min<long long>(np, 65535)), dim3(256), 0, g_stream,
                     (long long)n, (long long)ncols, in, out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// ---- gathers from the pair-first MO tensor ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) extract_pf_kernel(long long n, const double* __restrict__ Mp, long long p0, long long q0, long long r0, long long s0,
                                                        long long sp, long long sq, long long sr, long long ss, double* __restrict__ out) {
  const long long nrs = sr * ss;
  for (long long pq = blockIdx.x; pq < sp * sq; pq += gridDim.x) {
    const long long p = pq / sq, q = pq - p * sq;
    const double* src = Mp + pair_idx(p0 + p, q0 + q) * n * n + r0 * n + s0;
    double* dst = out + pq * nrs;
    for (long long t = threadIdx.x; t < nrs; t += blockDim.x) {
      const long long r = t / ss, s = t - r * ss;
      dst[t] = src[r * n + s];
    }
  }
}
int dev_extract_pf(int64_t n, const double* Mp, int64_t p0, int64_t q0, int64_t r0, int64_t s0, int64_t sp, int64_t sq, int64_t sr, int64_t ss, double* out) {
  REQUIRE_INIT();
  if (sp <= 0 || sq <= 0 || sr <= 0 || ss <= 0) return QEMB_OK;
  hipLaunchKernelGGL(extract_pf_kernel, dim3((unsigned)std::min<int64_t>(sp * sq, 1 << 20)), dim3(256), 0, g_stream, (long long)n, Mp, (long long)p0, (long long)q0,
                     (long long)r0, (long long)s0, (long long)sp, (long long)sq, (long long)sr, (long long)ss, out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// out[x][r][s][c] = T[P(r0+r,s0+s)][c0+c][x0+x]: one workgroup per (r,s) slab reads the sc x sx corner (rows of sx contiguous doubles)
__global__ void __launch_bounds__(256) extract_pf_t_kernel(long long n, const double* __restrict__ T, long long x0, long long r0, long long s0, long long c0,
                                                          long long sx, long long sr, long long ss, long long sc, double* __restrict__ out, long long slab) {
  for (long long rs = blockIdx.x; rs < sr * ss; rs += gridDim.x) {
    const long long r = rs / ss, s = rs - r * ss;
    const double* src = T + pair_idx(r0 + r, s0 + s) * slab + c0 * n + x0;
    for (long long t = threadIdx.x; t < sc * sx; t += blockDim.x) {
      const long long c = t / sx, x = t - c * sx;
      out[(x * sr * ss + rs) * sc + c] = src[c * n + x];
    }
  }
}
int dev_extract_pf_t(int64_t n, const double* T, int64_t x0, int64_t r0, int64_t s0, int64_t c0, int64_t sx, int64_t sr, int64_t ss, int64_t sc, double* out, int64_t slab) {
  REQUIRE_INIT();
  if (sx <= 0 || sr <= 0 || ss <= 0 || sc <= 0) return QEMB_OK;
  if (slab <= 0) slab = n * n;          // (a pair's slab holds all n rows; the factor route of mo_transform keeps only the first nf)
  if (c0 + sc > slab / n) { set_error("dev_extract_pf_t: rows beyond the slab"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(extract_pf_t_kernel, dim3((unsigned)std::min<int64_t>(sr * ss, 1 << 20)), dim3(256), 0, g_stream, (long long)n, T, (long long)x0, (long long)r0,
                     (long long)s0, (long long)c0, (long long)sx, (long long)sr, (long long)ss, (long long)sc, out, (long long)slab);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// one workgroup per output row P(a,b); for every c the two source runs (d = 0..c) are contiguous:
//   (ac|bd) = Mp[P(va,vc)][vb][vd],   (ad|bc) = (bc|ad) = Mp[P(vb,vc)][va][vd]
__global__ void __launch_bounds__(256) ladder_pack_vvvv_pf_kernel(long long n, long long o, const double* __restrict__ Mp,
                                                                 double* __restrict__ Vp, long long ldp, double* __restrict__ Vm, long long ldm) {
  const long long v = n - o, np = v * (v + 1) / 2, nm = v * (v - 1) / 2, n2 = n * n;
  for (long long ab = blockIdx.x; ab < np; ab += gridDim.x) {
    long long a, b; unpair_ge(ab, a, b);
    double* vp = Vp + ab * ldp;
    double* vm = (a > b) ? Vm + (a * (a - 1) / 2 + b) * ldm : nullptr;
    // four column pairs per thread and trip, all eight gathers requested before the first use: the two source runs of a (c,d) are short
    // (c + 1 doubles) pieces of different slabs, so the pass is bound by how many of them a CU keeps in flight (round 4: 3.1 -> see DESIGN)
    for (long long cd0 = threadIdx.x; cd0 < ldp; cd0 += 4 * (long long)blockDim.x) {
      double x[4], y[4]; long long cc[4], dd[4]; bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long cd = cd0 + u * (long long)blockDim.x;
        in[u] = cd < np;
        long long c = 0, d = 0;
        if (in[u]) unpair_ge(cd, c, d);
        cc[u] = c; dd[u] = d;
        x[u] = in[u] ? Mp[pair_idx(o + a, o + c) * n2 + (o + b) * n + (o + d)] : 0.0;
        y[u] = in[u] ? Mp[pair_idx(o + b, o + c) * n2 + (o + a) * n + (o + d)] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long cd = cd0 + u * (long long)blockDim.x;
        if (cd >= ldp) continue;
        if (!in[u]) { vp[cd] = 0.0; continue; }
        vp[cd] = x[u] + y[u];
        if (vm && cc[u] > dd[u]) vm[cc[u] * (cc[u] - 1) / 2 + dd[u]] = x[u] - y[u];
      }
    }
    if (vm) for (long long q = nm + threadIdx.x; q < ldm; q += blockDim.x) vm[q] = 0.0;
  }
}
int dev_ladder_pack_vvvv_pf(int64_t n, int64_t o, const double* Mp, double* Vp, int64_t ldp, double* Vm, int64_t ldm) {
  REQUIRE_INIT();
  const long long v = n - o, np = v * (v + 1) / 2;
  if (np <= 0) return QEMB_OK;
  hipLaunchKernelGGL(ladder_pack_vvvv_pf_kernel, dim3((unsigned)std::min<long long>(np, 1 << 20)), dim3(256), 0, g_stream, (long long)n, (long long)o, Mp, Vp, (long long)ldp, Vm, (long long)ldm);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ---- (+/-) packed ladder ------------------------------------------------------------------------------------
// one block row per (a >= b); threads run over P(c,d).  Reads M[a,c,b,d] (d contiguous) and M[a,d,b,c].
__global__ void __launch_bounds__(256) ladder_pack_vvvv_kernel(long long n, long long o, const double* __restrict__ M,
                                                              double* __restrict__ Vp, long long ldp, double* __restrict__ Vm, long long ldm) {
  const long long v = n - o, np = v * (v + 1) / 2;
  for (long long ab = blockIdx.x; ab < np; ab += gridDim.x) {
    long long a, b; unpair_ge(ab, a, b);
    const double* Ma = M + ((o + a) * n) * n * n + (o + b) * n;      // M[o+a, :, o+b, :]
    double* vp = Vp + ab * ldp;
    double* vm = (a > b) ? Vm + (a * (a - 1) / 2 + b) * ldm : nullptr;
    for (long long cd = threadIdx.x; cd < ldp; cd += blockDim.x) {
      if (cd >= np) { vp[cd] = 0.0; continue; }
      long long c, d; unpair_ge(cd, c, d);
      const double x = Ma[(o + c) * n * n + (o + d)], y = Ma[(o + d) * n * n + (o + c)];
      vp[cd] = x + y;
      if (vm && c > d) vm[c * (c - 1) / 2 + d] = x - y;
    }
    if (vm) { const long long nm = v * (v - 1) / 2; for (long long q = nm + threadIdx.x; q < ldm; q += blockDim.x) vm[q] = 0.0; }
  }
}
int dev_ladder_pack_vvvv(int64_t n, int64_t o, const double* M, double* Vp, int64_t ldp, double* Vm, int64_t ldm) {
  REQUIRE_INIT();
  const long long v = n - o, np = v * (v + 1) / 2;
  if (np <= 0) return QEMB_OK;
  hipLaunchKernelGGL(ladder_pack_vvvv_kernel, dim3((unsigned)std::min<long long>(np, 1 << 20)), dim3(256), 0, g_stream, (long long)n, (long long)o, M, Vp, (long long)ldp, Vm, (long long)ldm);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// (+/-) pair packing of the last two indices of v x v slabs through LDS tiles (round 3).  The element-wise kernels below read in[c][d] along d
// but in[d][c] with a stride of v doubles -- one cache line per lane -- and spend a double-precision square root per element on the pair
// index: 2.6 TB/s (pack_pm_cols), 3.8 TB/s (ladder_pack_tau).  Here a workgroup takes one pair of 32 x 32 tiles (tc >= td) of a slab: tile
// (tc, td) and its mirror (td, tc) are both read along rows, the mirror is read transposed out of LDS, and each thread row writes 32
// consecutive packed entries c(c+1)/2 + d.  MODE 0: Op = x + y, Om = x - y for every row.  MODE 1 (tau): the slab of packed row ij is
// (i*o + j), Op = w (x + y) with w = 1/2 (1/4 on c = d), Om = (x - y)/2 only for i > j (row Q(i,j)).
template <int MODE>
__global__ void __launch_bounds__(256) pack_pm_tiled_kernel(long long rows, long long o, long long v, const double* __restrict__ in, double* __restrict__ Op,
                                                            long long ldp, double* __restrict__ Om, long long ldm) {
  __shared__ double tB[32][33];       // only the mirror tile goes through LDS; the straight tile stays in the registers of the threads that write it
  const long long np = v * (v + 1) / 2, nm = v * (v - 1) / 2;
  const uint3 LB = xcd_logical_block(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
  long long tc, td; unpair_ge((long long)LB.x, tc, td);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bool diag = (tc == td);
  for (long long r = LB.y; r < rows; r += gridDim.y) {
    const double* t; double* tp = Op + r * ldp; double* tm;
    if (MODE == 1) {
      long long i, j; unpair_ge(r, i, j);
      t = in + (i * o + j) * v * v;
      tm = (i > j) ? Om + (i * (i - 1) / 2 + j) * ldm : nullptr;
    } else {
      t = in + r * v * v; tm = Om + r * ldm;
    }
    double xa[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = ty + 8 * k;
      const long long c = tc * 32 + rr, d = td * 32 + tx;
      xa[k] = (c < v && d < v) ? t[c * v + d] : 0.0;
      if (!diag) {
        const long long d2 = td * 32 + rr, c2 = tc * 32 + tx;
        tB[rr][tx] = (d2 < v && c2 < v) ? t[d2 * v + c2] : 0.0;
      } else {
        tB[rr][tx] = xa[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = ty + 8 * k;
      const long long c = tc * 32 + cc, d = td * 32 + tx;
      if (c < v && d <= c) {
        const double x = xa[k], y = tB[tx][cc];
        if (MODE == 1) {
          tp[c * (c + 1) / 2 + d] = (c == d) ? 0.25 * (x + y) : 0.5 * (x + y);
          if (tm && c > d) tm[c * (c - 1) / 2 + d] = 0.5 * (x - y);
        } else {
          tp[c * (c + 1) / 2 + d] = x + y;
          if (c > d) tm[c * (c - 1) / 2 + d] = x - y;
        }
      }
    }
    if (LB.x == 0) {      // padding columns of the row (leading dimensions are rounded up to even)
      for (long long q = np + threadIdx.x; q < ldp; q += 256) tp[q] = 0.0;
      if (tm) for (long long q = nm + threadIdx.x; q < ldm; q += 256) tm[q] = 0.0;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void ladder_pack_tau_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, const double* __restrict__ tau,
                                                             double* __restrict__ Tp, long long ldp, double* __restrict__ Tm, long long ldm) {
  const long long ij = BID.x, np = v * (v + 1) / 2, nm = v * (v - 1) / 2;
  long long i, j; unpair_ge(ij, i, j);
  const double* t = tau + (i * o + j) * v * v;
  double* tp = Tp + ij * ldp;
  double* tm = (i > j) ? Tm + (i * (i - 1) / 2 + j) * ldm : nullptr;
  // BID.y: a slice of the packed (c,d) range -- one workgroup per pair (ij) alone is 210 workgroups at n_occ = 20, fewer than the chip has CUs
  const long long chunk = (ldp + GDIM.y - 1) / GDIM.y, cd0 = BID.y * chunk, cd1 = (cd0 + chunk < ldp) ? cd0 + chunk : ldp;
  for (long long cd = cd0 + threadIdx.x; cd < cd1; cd += blockDim.x) {
    if (cd >= np) { tp[cd] = 0.0; continue; }
    long long c, d; unpair_ge(cd, c, d);
    const double x = t[c * v + d], y = t[d * v + c];
    tp[cd] = (c == d) ? 0.25 * (x + y) : 0.5 * (x + y);
    if (tm && c > d) tm[c * (c - 1) / 2 + d] = 0.5 * (x - y);
  }
  if (tm && BID.y == 0) for (long long q = nm + threadIdx.x; q < ldm; q += blockDim.x) tm[q] = 0.0;
}
__global__ void __launch_bounds__(256) ladder_pack_tau_kernel(long long o, long long v, const double* __restrict__ tau,
                                                             double* __restrict__ Tp, long long ldp, double* __restrict__ Tm, long long ldm) { ladder_pack_tau_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, tau, Tp, ldp, Tm, ldm); }
int dev_ladder_pack_tau(int64_t o, int64_t v, const double* tau, double* Tp, int64_t ldp, double* Tm, int64_t ldm) {
  REQUIRE_INIT();
  const long long npo = o * (o + 1) / 2;
  if (npo <= 0 || v <= 0) return QEMB_OK;
  if (v >= 32) {
    const long long nt = (v + 31) / 32;
    hipLaunchKernelGGL(pack_pm_tiled_kernel<1>, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)std::min<long long>(npo, 65535)), dim3(256), 0, g_stream,
                       npo, (long long)o, (long long)v, tau, Tp, (long long)ldp, Tm, (long long)ldm);
    HIP_TRY(hipGetLastError());
    return QEMB_OK;
  }
  const unsigned slices = (unsigned)std::max<long long>(1, std::min<long long>(16, ldp / 2048));
  hipLaunchKernelGGL(ladder_pack_tau_kernel, dim3((unsigned)npo, slices), dim3(256), 0, g_stream, (long long)o, (long long)v, tau, Tp, (long long)ldp, Tm, (long long)ldm);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) pack_pm_cols_kernel(long long rows, long long v, const double* __restrict__ in, double* __restrict__ Op, long long ldp,
                                                          double* __restrict__ Om, long long ldm) {
  const long long np = v * (v + 1) / 2, nm = v * (v - 1) / 2;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const double* t = in + r * v * v;
    double* tp = Op + r * ldp;
    double* tm = Om + r * ldm;
    for (long long cd = threadIdx.x; cd < ldp; cd += blockDim.x) {
      if (cd >= np) { tp[cd] = 0.0; continue; }
      long long c, d; unpair_ge(cd, c, d);
      const double x = t[c * v + d], y = t[d * v + c];
      tp[cd] = x + y;
      if (c > d) tm[c * (c - 1) / 2 + d] = x - y;
    }
    for (long long q = nm + threadIdx.x; q < ldm; q += blockDim.x) tm[q] = 0.0;
  }
}
int dev_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm) {
  REQUIRE_INIT();
  if (rows <= 0 || v <= 0) return QEMB_OK;
  if (v >= 32) {
    const long long nt = (v + 31) / 32;
    hipLaunchKernelGGL(pack_pm_tiled_kernel<0>, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)std::min<int64_t>(rows, 16384)), dim3(256), 0, g_stream,
                       (long long)rows, 0LL, (long long)v, in, Op, (long long)ldp, Om, (long long)ldm);
    HIP_TRY(hipGetLastError());
    return QEMB_OK;
  }
  hipLaunchKernelGGL(pack_pm_cols_kernel, dim3((unsigned)std::min<int64_t>(rows, 1 << 20)), dim3(256), 0, g_stream, (long long)rows, (long long)v, in, Op, (long long)ldp, Om, (long long)ldm);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__device__ __forceinline__ void scatter_pm_rows_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long ncols, const double* __restrict__ Xp, const double* __restrict__ Xm, double* __restrict__ out, const double* __restrict__ add,
                                                            int Sp, long long strideP, int Sm, long long strideM) {
  const long long ij = BID.y;
  long long i, j; unpair_ge(ij, i, j);
  const double* xp = Xp + ij * ncols;
  const double* xm = (i > j) ? Xm + (i * (i - 1) / 2 + j) * ncols : nullptr;
  double* oij = out + (i * o + j) * ncols;
  double* oji = out + (j * o + i) * ncols;
  const double* aij = add ? add + (i * o + j) * ncols : nullptr;
  const double* aji = add ? add + (j * o + i) * ncols : nullptr;
  for (long long c = (long long)BID.x * blockDim.x + threadIdx.x; c < ncols; c += (long long)GDIM.x * blockDim.x) {
    const double p = slab_sum(xp + c, Sp, strideP);         // (split-K slabs: the sum the reduction pass would have formed)
    if (xm) { const double m = slab_sum(xm + c, Sm, strideM); oij[c] = add ? aij[c] + (p + m) : p + m; oji[c] = add ? aji[c] + (p - m) : p - m; }
    else oij[c] = add ? aij[c] + p : p;
  }
}
__global__ void __launch_bounds__(256) scatter_pm_rows_kernel(long long o, long long ncols, const double* __restrict__ Xp, const double* __restrict__ Xm, double* __restrict__ out, const double* __restrict__ add, int Sp, long long strideP, int Sm, long long strideM) { scatter_pm_rows_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, ncols, Xp, Xm, out, add, Sp, strideP, Sm, strideM); }
int dev_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out, const double* add, int Sp, int64_t strideP, int Sm, int64_t strideM) {
  REQUIRE_INIT();
  const long long npo = o * (o + 1) / 2;
  if (npo <= 0 || ncols <= 0) return QEMB_OK;
  if (npo > 65535) { set_error("dev_scatter_pm_rows: too many pairs"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(scatter_pm_rows_kernel, dim3((unsigned)std::min<long long>((ncols + 255) / 256, 64), (unsigned)npo), dim3(256), 0, g_stream, (long long)o, (long long)ncols, Xp, Xm, out, add, std::max(Sp, 1), (long long)strideP, std::max(Sm, 1), (long long)strideM);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// grid (32 x 32 tiles of (a,b), o*o): U[j,i,b,a] is read with a fastest (coalesced) and transposed through LDS
__device__ __forceinline__ void ccsd_finish_t2_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, double* __restrict__ t2n, const double* __restrict__ U,
                                                            const double* __restrict__ OV, const double* __restrict__ eo, const double* __restrict__ ev) {
  __shared__ double tile[32][33];
  const long long ij = BID.y, i = ij / o, j = ij - i * o;
  const long long nt = (v + 31) / 32;
  const long long ta = BID.x / nt, tb = BID.x - ta * nt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* Uji = U + (j * o + i) * v * v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {      // tile[bb][aa] = U[j,i,b,a]
    const int bb = ty + 8 * k;
    const long long b = tb * 32 + bb, a = ta * 32 + tx;
    tile[bb][tx] = (a < v && b < v) ? Uji[b * v + a] : 0.0;
  }
  __syncthreads();
  const double eij = eo[i] + eo[j];
  const long long base = ij * v * v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int aa = ty + 8 * k;
    const long long a = ta * 32 + aa, b = tb * 32 + tx;
    if (a < v && b < v) {
      const long long idx = base + a * v + b;
      t2n[idx] = (t2n[idx] + OV[idx] + U[idx] + tile[tx][aa]) / (eij - ev[a] - ev[b]);
    }
  }
}
__global__ void __launch_bounds__(256) ccsd_finish_t2_kernel(long long o, long long v, double* __restrict__ t2n, const double* __restrict__ U,
                                                            const double* __restrict__ OV, const double* __restrict__ eo, const double* __restrict__ ev) { ccsd_finish_t2_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, t2n, U, OV, eo, ev); }
int dev_ccsd_finish_t2(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* eo, const double* ev) {
  REQUIRE_INIT();
  if (o <= 0 || v <= 0) return QEMB_OK;
  if (o * o > 65535) { set_error("dev_ccsd_finish_t2: too many occupied pairs"); return QEMB_ERR_ARG; }
  const long long nt = (v + 31) / 32;
  hipLaunchKernelGGL(ccsd_finish_t2_kernel, dim3((unsigned)(nt * nt), (unsigned)(o * o)), dim3(256), 0, g_stream, (long long)o, (long long)v, t2n, U, OV, eo, ev);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// finish_t2 with the two ring products taken where the GEMMs left them ([o][v][o][v]) and every (i >= j) PAIR of tiles handled once: the tile (ta,tb) of
// t2n[i,j] and the tile (tb,ta) of t2n[j,i] are transposes of each other, so one workgroup reads the ten operand tiles, forms the 32 x 32 result once and
// stores it both ways (the second through LDS).  grid (nt * nt tiles, npair(o)); a diagonal pair i == j does its tiles twice over (ta,tb) and (tb,ta): same values.
__device__ __forceinline__ void ccsd_finish_t2_rings_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, double* __restrict__ t2n, const double* __restrict__ U,
                                                                   const double* __restrict__ OV, const double* __restrict__ RS, const double* __restrict__ M,
                                                                   const double* __restrict__ eo, const double* __restrict__ ev, double* __restrict__ t1n) {
  __shared__ double tile[32][33];
  const uint3 LB = xcd_logical_block(BID, GDIM);
  long long i, j; unpair_ge((long long)LB.y, i, j);
  const long long nt = (v + 31) / 32;
  const long long ta = LB.x / nt, tb = LB.x - ta * nt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long vv = v * v, ov = o * v;
  const double* Uij = U + (i * o + j) * vv;
  const double* Uji = U + (j * o + i) * vv;
  // [o][v][o][v] operands: element [i,x,j,y] at (i * v + x) * ov + j * v + y
  const double* RSij = RS + i * v * ov + j * v;
  const double* RSji = RS + j * v * ov + i * v;
  const double* Mij = M + i * v * ov + j * v;
  const double* Mji = M + j * v * ov + i * v;
  // transposed contributions, summed before the transpose: tile[bb][aa] = -M[i,b,j,a] + U[j,i,b,a] + RS[j,b,i,a] - 1/2 M[j,b,i,a]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int bb = ty + 8 * k;
    const long long b = tb * 32 + bb, a = ta * 32 + tx;
    tile[bb][tx] = (a < v && b < v) ? ((Uji[b * v + a] + RSji[b * ov + a]) - 0.5 * Mji[b * ov + a]) - Mij[b * ov + a] : 0.0;
  }
  __syncthreads();
  const double eij = eo[i] + eo[j];
  double res[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int aa = ty + 8 * k;
    const long long a = ta * 32 + aa, b = tb * 32 + tx;
    res[k] = 0.0;
    if (a < v && b < v) {
      const long long idx = (i * o + j) * vv + a * v + b;
      const double straight = ((Uij[a * v + b] + RSij[a * ov + b]) - 0.5 * Mij[a * ov + b]) - Mji[a * ov + b];
      res[k] = (t2n[idx] + OV[idx] + straight + tile[tx][aa]) / (eij - ev[a] - ev[b]);
      t2n[idx] = res[k];
    }
  }
  if (i != j) {        // the partner tile t2n[j,i,b,a] = t2n[i,j,a,b], written in rows of b
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = res[k];      // tile[aa][bb]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int bb = ty + 8 * k;
      const long long b = tb * 32 + bb, a = ta * 32 + tx;
      if (a < v && b < v) t2n[(j * o + i) * vv + b * v + a] = tile[tx][bb];
    }
  }
  if (t1n && LB.x == 0 && LB.y == 0)
    for (long long t = threadIdx.x; t < ov; t += blockDim.x) t1n[t] /= eo[t / v] - ev[t % v];
}
__global__ void __launch_bounds__(256) ccsd_finish_t2_rings_kernel(long long o, long long v, double* __restrict__ t2n, const double* __restrict__ U, const double* __restrict__ OV,
                                                                  const double* __restrict__ RS, const double* __restrict__ M, const double* __restrict__ eo,
                                                                  const double* __restrict__ ev, double* __restrict__ t1n) {
  ccsd_finish_t2_rings_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, t2n, U, OV, RS, M, eo, ev, t1n);
}
int dev_ccsd_finish_t2_rings(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* RS, const double* M, const double* eo, const double* ev, double* t1n) {
  REQUIRE_INIT();
  if (o <= 0 || v <= 0) return QEMB_OK;
  const long long npo = o * (o + 1) / 2, nt = (v + 31) / 32;
  if (npo > 65535) { set_error("dev_ccsd_finish_t2_rings: too many occupied pairs"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(ccsd_finish_t2_rings_kernel, dim3((unsigned)(nt * nt), (unsigned)npo), dim3(256), 0, g_stream, (long long)o, (long long)v, t2n, U, OV, RS, M, eo, ev, t1n);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// grid (lower-triangle 32 x 32 tiles of (a,b), npair(o)): the tile of R+/R- is staged through LDS so that both the [a][b] image
// and its mirror [b][a] are updated in 256-byte runs, for t2[i,j] and t2[j,i].
__device__ __forceinline__ void ladder_scatter_pm_kernel_body(const uint3 BID, const uint3 GDIM, long long o, long long v, const double* __restrict__ Rp, long long ldp,
                                                               const double* __restrict__ Rm, long long ldm, double* __restrict__ t2,
                                                               const double* __restrict__ Hp, const double* __restrict__ Hm, int assign,
                                                               int Sp, long long strideP, int Sm, long long strideM, long long ldhp, long long ldhm) {
  __shared__ double sp[32][33], sm[32][33];
  const uint3 LB = xcd_logical_block(BID, GDIM);
  const long long ij = LB.y;
  long long i, j; unpair_ge(ij, i, j);
  long long t = LB.x, ta, tb; unpair_ge(t, ta, tb);            // tile row >= tile column
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* rp = Rp + ij * ldp;
  const double* rm = (i > j) ? Rm + (i * (i - 1) / 2 + j) * ldm : nullptr;
  const double* hp = Hp ? Hp + ij * ldhp : nullptr;
  const double* hm = (Hm && i > j) ? Hm + (i * (i - 1) / 2 + j) * ldhm : nullptr;
  double* tij = t2 + (i * o + j) * v * v;
  double* tji = t2 + (j * o + i) * v * v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int aa = ty + 8 * k;
    const long long a = ta * 32 + aa, b = tb * 32 + tx;
    double p = 0.0, m = 0.0;
    if (a < v && b <= a) {
      p = slab_sum(rp + a * (a + 1) / 2 + b, Sp, strideP);       // (split-K slabs: the sum the reduction pass would have formed)
      if (hp) p += (a == b ? 2.0 : 1.0) * hp[a * (a + 1) / 2 + b];
      if (rm && b < a) { m = slab_sum(rm + a * (a - 1) / 2 + b, Sm, strideM); if (hm) m += hm[a * (a - 1) / 2 + b]; }
    }
    sp[aa][tx] = p; sm[aa][tx] = m;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int aa = ty + 8 * k;
    {  // image [a][b], b <= a
      const long long a = ta * 32 + aa, b = tb * 32 + tx;
      if (a < v && b <= a) {
        const double p = sp[aa][tx], m = sm[aa][tx];
        if (assign) { tij[a * v + b] = p + m; if (i != j) tji[a * v + b] = p - m; }
        else { tij[a * v + b] += p + m; if (i != j) tji[a * v + b] += p - m; }
      }
    }
    {  // mirror [b][a], b < a: destination row r = tb*32 + aa, column c = ta*32 + tx holds the element (a = c, b = r)
      const long long r = tb * 32 + aa, c = ta * 32 + tx;
      if (c < v && r < c) {
        const double p = sp[tx][aa], m = sm[tx][aa];
        if (assign) { tij[r * v + c] = p - m; if (i != j) tji[r * v + c] = p + m; }
        else { tij[r * v + c] += p - m; if (i != j) tji[r * v + c] += p + m; }
      }
    }
  }
}
__global__ void __launch_bounds__(256) ladder_scatter_pm_kernel(long long o, long long v, const double* __restrict__ Rp, long long ldp,
                                                               const double* __restrict__ Rm, long long ldm, double* __restrict__ t2,
                                                               const double* __restrict__ Hp, const double* __restrict__ Hm, int assign,
                                                               int Sp, long long strideP, int Sm, long long strideM, long long ldhp, long long ldhm) { ladder_scatter_pm_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, Rp, ldp, Rm, ldm, t2, Hp, Hm, assign, Sp, strideP, Sm, strideM, ldhp, ldhm); }
int dev_ladder_scatter_pm2(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, const double* Hp, const double* Hm,
                           int assign, double* t2, int Sp, int64_t strideP, int Sm, int64_t strideM, int64_t ldhp, int64_t ldhm) {
  REQUIRE_INIT();
  const long long npo = o * (o + 1) / 2;
  if (npo <= 0 || v <= 0) return QEMB_OK;
  if (npo > 65535) { set_error("dev_ladder_scatter_pm: too many pairs"); return QEMB_ERR_ARG; }
  const long long nt = (v + 31) / 32;
  hipLaunchKernelGGL(ladder_scatter_pm_kernel, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)npo), dim3(256), 0, g_stream, (long long)o, (long long)v, Rp, (long long)ldp, Rm, (long long)ldm, t2,
                     Hp, Hm, assign, std::max(Sp, 1), (long long)strideP, std::max(Sm, 1), (long long)strideM, (long long)(ldhp ? ldhp : ldp), (long long)(ldhm ? ldhm : ldm));
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_ladder_scatter_pm(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, double* t2) {
  return dev_ladder_scatter_pm2(o, v, Rp, ldp, Rm, ldm, nullptr, nullptr, 0, t2);
}
// one thread per (P(ij), P(kl)) / (Q(ij), Q(kl)) entry of the packed images of W[k,l,i,j]; W(k,l,i,j) is a functor: a stored tensor (pack_w_pm) or the
// four-term sum that IS the Woooo intermediate (pack_w_pm_sum: Wt[i,j,k,l] + X[i,j,k,l] + At[j,i,k,l] + At[i,j,l,k], added in that order)
// grid (ceil(lda_p / 256), npair(o)): row P(ij) per BID.y, one thread per column P(kl).  W(k,l,i,j) and W(k,l,j,i) are read once and serve both images
// (the (-) entry Q(ij),Q(kl) exists when i > j and k > l); the padding columns are zeroed by the threads past npair(o) / by the first thread of the row.
template <class WF>
__device__ __forceinline__ void pack_w_pm_any(const uint3 BID, long long o, WF W, double* __restrict__ Ap, long long lda_p, double* __restrict__ Am, long long lda_m) {
  const long long npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2;
  long long i, j; unpair_ge((long long)BID.y, i, j);
  const long long kl = (long long)BID.x * blockDim.x + threadIdx.x;
  if (kl >= lda_p) return;
  double* ap = Ap + (long long)BID.y * lda_p;
  double* am = (Am && i > j) ? Am + (i * (i - 1) / 2 + j) * lda_m : nullptr;
  if (kl >= npo) { ap[kl] = 0.0; return; }
  long long k, l; unpair_ge(kl, k, l);
  const double wa = W(k, l, i, j), wb = (k == l) ? 0.0 : W(k, l, j, i);
  ap[kl] = (k == l) ? wa : wa + wb;
  if (am) {
    if (k > l) am[k * (k - 1) / 2 + l] = wa - wb;
    if (kl == 0) for (long long c = nmo; c < lda_m; ++c) am[c] = 0.0;
  }
}
__device__ __forceinline__ void pack_w_pm_kernel_body(const uint3 BID, const uint3 GDIM, long long o, const double* __restrict__ W, double* __restrict__ Ap, long long lda_p,
                                                        double* __restrict__ Am, long long lda_m) {
  pack_w_pm_any(BID, o, [=](long long k, long long l, long long i, long long j) { return W[((k * o + l) * o + i) * o + j]; }, Ap, lda_p, Am, lda_m);
}
__global__ void __launch_bounds__(256) pack_w_pm_kernel(long long o, const double* __restrict__ W, double* __restrict__ Ap, long long lda_p,
                                                        double* __restrict__ Am, long long lda_m) { pack_w_pm_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, W, Ap, lda_p, Am, lda_m); }
__device__ __forceinline__ void pack_w_pm_sum_kernel_body(const uint3 BID, const uint3 GDIM, long long o, const double* __restrict__ Wp, const double* __restrict__ X, const double* __restrict__ O1,      // (Wp = Wt, O1 = At of dev_ops.h)
                                                            double* __restrict__ Ap, long long lda_p, double* __restrict__ Am, long long lda_m) {
  // every operand is addressed [row pair][column pair]: a row of the packed images reads the (i,j) and (j,i) blocks of o^2 contiguous doubles of each (round 5:
  // with W stored [k,l,i,j] and O1 stored [l,j,k,i] neighbouring threads read o^2 doubles apart -- 0.25 TB/s on the n_occ ~ 28 fragments of octane BE3)
  pack_w_pm_any(BID, o, [=](long long k, long long l, long long i, long long j) {
    return ((Wp[((i * o + j) * o + k) * o + l] + X[((i * o + j) * o + k) * o + l]) + O1[((j * o + i) * o + k) * o + l]) + O1[((i * o + j) * o + l) * o + k]; }, Ap, lda_p, Am, lda_m);
}
__global__ void __launch_bounds__(256) pack_w_pm_sum_kernel(long long o, const double* __restrict__ Wp, const double* __restrict__ X, const double* __restrict__ O1,
                                                            double* __restrict__ Ap, long long lda_p, double* __restrict__ Am, long long lda_m) {
  pack_w_pm_sum_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, Wp, X, O1, Ap, lda_p, Am, lda_m);
}
// one wave per output (k,i), lanes over l (then strides of 64): a thread per output walks l in a chain of L2 round trips (14.7 us at n_occ = 21)
__device__ __forceinline__ double wave_sum_fwd(double v);
__device__ __forceinline__ void foo_from_x_kernel_body(const uint3 BID, const uint3 GDIM, long long o, const double* __restrict__ X, double* __restrict__ F) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)BID.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= o * o) return;
  const long long k = t / o, i = t - k * o;
  double s = 0.0;
  for (long long l = lane; l < o; l += 64) s += 2.0 * X[((i * o + l) * o + k) * o + l] - X[((l * o + i) * o + k) * o + l];
  s = wave_sum_fwd(s);
  if (lane == 0) F[t] = s;
}
__global__ void __launch_bounds__(256) foo_from_x_kernel(long long o, const double* __restrict__ X, double* __restrict__ F) { foo_from_x_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, X, F); }
int dev_foo_from_x(int64_t o, const double* X, double* F) {
  REQUIRE_INIT();
  if (o <= 0) return QEMB_OK;
  hipLaunchKernelGGL(foo_from_x_kernel, dim3((unsigned)((o * o + 3) / 4)), dim3(256), 0, g_stream, (long long)o, X, F);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_pack_w_pm_sum(int64_t o, const double* Wp, const double* X, const double* O1, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) {
  REQUIRE_INIT();
  if (o <= 0) return QEMB_OK;
  const long long npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2;
  if (lda_p < npo || (nmo > 0 && lda_m < nmo)) { set_error("dev_pack_w_pm_sum: leading dimension too small"); return QEMB_ERR_ARG; }
  if (npo > 65535) { set_error("dev_pack_w_pm_sum: too many pairs"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(pack_w_pm_sum_kernel, dim3((unsigned)((lda_p + 255) / 256), (unsigned)npo), dim3(256), 0, g_stream, (long long)o, Wp, X, O1, Ap, (long long)lda_p, nmo > 0 ? Am : nullptr, (long long)lda_m);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_pack_w_pm(int64_t o, const double* W, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) {
  REQUIRE_INIT();
  if (o <= 0) return QEMB_OK;
  const long long npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2;
  if (lda_p < npo || (nmo > 0 && lda_m < nmo)) { set_error("dev_pack_w_pm: leading dimension too small"); return QEMB_ERR_ARG; }
  if (npo > 65535) { set_error("dev_pack_w_pm: too many pairs"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(pack_w_pm_kernel, dim3((unsigned)((lda_p + 255) / 256), (unsigned)npo), dim3(256), 0, g_stream, (long long)o, W, Ap, (long long)lda_p, nmo > 0 ? Am : nullptr, (long long)lda_m);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// reductions (deterministic: fixed grid, fixed tree)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_fwd(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
template <bool MAX>
__device__ __forceinline__ double block_reduce(double v, double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = MAX ? wave_max(v) : wave_sum(v);
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) { r = sh[0]; for (int k = 1; k < nw; ++k) r = MAX ? fmax(r, sh[k]) : r + sh[k]; }
  __syncthreads();
  return r;  // valid on thread 0
}
template <bool MAX>
__global__ void __launch_bounds__(256) reduce_stage1(long long n, const double* x, const double* y, double* partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    acc = MAX ? fmax(acc, fabs(x[i])) : acc + x[i] * y[i];
  const double r = block_reduce<MAX>(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
template <bool MAX>
__global__ void __launch_bounds__(256) reduce_stage2(int np, const double* partial, double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc = MAX ? fmax(acc, partial[i]) : acc + partial[i];
  const double r = block_reduce<MAX>(acc, sh);
  if (threadIdx.x == 0) out[0] = r;
}
int dev_dot(int64_t n, const double* x, const double* y, double* out_dev) {
  REQUIRE_INIT();
  const int np = (int)std::max<int64_t>(1, std::min<int64_t>((n + 1023) / 1024, NPART));
  hipLaunchKernelGGL(reduce_stage1<false>, dim3(np), dim3(256), 0, g_stream, (long long)n, x, y, g_partials);
  hipLaunchKernelGGL(reduce_stage2<false>, dim3(1), dim3(256), 0, g_stream, np, g_partials, out_dev);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// out[j] = <x, ys[j]>, j < m <= 8, reading x once.  Same partition and summation order as dev_dot, so each result is
// bit-identical to the single dot product.
struct DotManyK { const double* y[8]; int m; };
__global__ void __launch_bounds__(256) dot_many_stage1(long long n, const double* __restrict__ x, DotManyK k, double* __restrict__ partial, int np) {
  __shared__ double sh[4];
  double acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double xi = x[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < k.m) acc[j] += xi * k.y[j][i];
  }
  for (int j = 0; j < k.m; ++j) {
    const double r = block_reduce<false>(acc[j], sh);
    if (threadIdx.x == 0) partial[(long long)j * np + blockIdx.x] = r;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) dot_many_stage2(int np, const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double sh[4];
  const int j = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partial[(long long)j * np + i];
  const double r = block_reduce<false>(acc, sh);
  if (threadIdx.x == 0) out[j] = r;
}
int dev_dot_many(int64_t n, const double* x, int m, const double* const* ys, double* out_dev) {
  REQUIRE_INIT();
  if (m <= 0) return QEMB_OK;
  if (m > 8) { set_error("dev_dot_many: at most 8 vectors"); return QEMB_ERR_ARG; }
  const int np = (int)std::max<int64_t>(1, std::min<int64_t>((n + 1023) / 1024, NPART));
  DotManyK k{};
  k.m = m;
  for (int j = 0; j < m; ++j) k.y[j] = ys[j];
  hipLaunchKernelGGL(dot_many_stage1, dim3(np), dim3(256), 0, g_stream, (long long)n, x, k, g_partials, np);
  hipLaunchKernelGGL(dot_many_stage2, dim3(m), dim3(256), 0, g_stream, np, (const double*)g_partials, out_dev);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// ---- single-launch reductions: every workgroup leaves its partial sums, the LAST one to finish (device-scope counter) adds them up in the
// fixed order of the two-stage reductions above and writes the result to device memory and to a pinned host word -- no second kernel, no copy
// node.  The counter lives behind the context's partials and is left at zero.
// finish: 2 = thread 0 alone fences (it is the only writer of the partials; its acquire fence empties the caches the whole workgroup reads through),
// 0 = do not finish here (a second small kernel does, QEMB_POST_FINISH=0).  (Every THREAD fencing, the textbook form, costs ~90 us per launch
// here -- each fence is an L2 write-back -- and made the fused launches slower than the five they replace.)
__device__ __forceinline__ bool last_workgroup(unsigned* counter, unsigned total, int finish) {
  __shared__ unsigned ticket;
  if (finish == 0) return false;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                 // this workgroup's partials are visible device wide before its ticket is
    ticket = atomicAdd(counter, 1u);
    if (ticket == total - 1) __threadfence();        // ... and the last one sees everybody's
  }
  __syncthreads();
  return ticket == total - 1;
}
__device__ __forceinline__ void publish_flag(unsigned long long* flag_host, unsigned long long seq) {   // one thread, after its result stores
  if (!flag_host) return;
  __threadfence_system();
  __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int post_finish_mode() {
  static const int m = [] { const char* e = std::getenv("QEMB_POST_FINISH"); return (e && e[0] == '2') ? 2 : 0; }();
  return m;
}
// second kernel of the QEMB_POST_FINISH=0 form: out[j] = sum of the np partials of reduction j, j < m, to device and pinned host memory
__device__ __forceinline__ void finish_partials_kernel_body(const uint3 BID, const uint3 GDIM, int m, int np, const double* __restrict__ partial, double* out_dev, double* out_host,
                                                              unsigned long long* flag_host, unsigned long long seq) {
  __shared__ double sh[4];
  double acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {        // the loads of all m reductions are in flight together; same order of additions as one after the other
    acc[j] = 0.0;
    if (j < m) for (int i = threadIdx.x; i < np; i += blockDim.x) acc[j] += partial[(long long)j * np + i];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j >= m) break;
    const double r = block_reduce<false>(acc[j], sh);
    if (threadIdx.x == 0) { out_dev[j] = r; out_host[j] = r; }
  }
  if (threadIdx.x == 0) publish_flag(flag_host, seq);
}
__global__ void __launch_bounds__(256) finish_partials_kernel(int m, int np, const double* __restrict__ partial, double* out_dev, double* out_host,
                                                              unsigned long long* flag_host, unsigned long long seq) {
  finish_partials_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), m, np, partial, out_dev, out_host, flag_host, seq);
}
int dev_wait_flag(const void* flag_host, unsigned long long seq) {
  REQUIRE_INIT();
  const unsigned long long* p = (const unsigned long long*)flag_host;
  for (unsigned long spins = 1;; ++spins) {
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) return QEMB_OK;
    if ((spins & 0xffff) == 0) {       // now and then: has the stream drained (or failed) without the word arriving?
      const hipError_t q = hipStreamQuery(g_stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) return QEMB_OK;
        set_error("dev_wait_flag: the stream has drained and the awaited word was never written");
        return QEMB_ERR_DEVICE;
      }
      if (q != hipErrorNotReady) HIP_TRY(q);
    }
    __builtin_ia32_pause();
    if ((spins & 0x7ff) == 0) std::this_thread::yield();      // several ranks x streams may share fewer cores than there are waiting threads
  }
}
struct DiisPushK { const double* y[8]; int m, self; };
struct ExtrapK { const double* x[8]; double c[8]; int n, write_amp; };
// ---- collected launches: between dev_batch_begin and dev_batch_flush the calling thread's dev_diis_push / dev_ccsd_extrapolate_energy calls are not launched but
// kept; the flush issues them as grouped launches (one per kernel for up to eight members) on the stream of the context bound THEN -- the lock-step sweep ends an
// iteration of six fragments with four launches instead of twenty-four.  Every collected call gets its own partial-sum region.
struct BatchEntry {
  int kind;                 // 0 / 1: diis_push scalar / 16-byte; 2 / 3: extrapolate scalar / 16-byte
  unsigned gx, gy; size_t lds;
  // arguments, in kernel order (pointers into this entry are what the grouped launch reads)
  long long n; const double* trial; const double* prev; double* e; double* xcopy; DiisPushK pk;
  int o, v, rows; ExtrapK ek; double* amp; const double* L; double* tau;
  double* partial; unsigned* counter; double* out_dev; double* out_host; int finish; unsigned long long* flag; unsigned long long seq;
  int fin_m, fin_np;
  void* params[16]; void* fparams[8];
};
struct BatchCollector { bool on = false; std::vector<BatchEntry> entries; double* scratch = nullptr; size_t scratch_regions = 0; };
static thread_local BatchCollector t_batch;
static constexpr size_t BATCH_REGION = 8 * NPART + 8;      // doubles per collected call
static double* batch_region(size_t k) {
  if (k >= t_batch.scratch_regions) {
    const size_t want = std::max<size_t>(16, 2 * (k + 1));
    double* p = nullptr;
    if (hipMalloc((void**)&p, want * BATCH_REGION * sizeof(double)) != hipSuccess) return nullptr;
    // (the old block, if any, may still be read by launches in flight: it is small and kept until the process ends)
    t_batch.scratch = p; t_batch.scratch_regions = want;
  }
  return t_batch.scratch + k * BATCH_REGION;
}
// VEC2: two consecutive elements per thread and step through 16-byte accesses (n even, every vector 16-byte aligned).  All loads of a step are issued
// before its stores: the vectors may alias each other (xcopy == prev in the first iteration), so a load written after a store would wait for it -- and
// with it for the loads that fed the store: two dependent round trips to HBM per step instead of one.
template <bool VEC2>
__device__ __forceinline__ void diis_push_kernel_body(const uint3 BID, const uint3 GDIM, long long n, const double* trial, const double* prev, double* e, double* xcopy,
                                                        DiisPushK k, double* partial, unsigned* counter, double* row_dev, double* row_host, int finish,
                                                        unsigned long long* flag_host, unsigned long long seq) {
  __shared__ double sh[4];
  const int np = (int)GDIM.x;
  constexpr int W = VEC2 ? 2 : 1;
  double acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0;
  const long long nw = n / W;
  for (long long i = (long long)BID.x * blockDim.x + threadIdx.x; i < nw; i += (long long)np * blockDim.x) {
    double t[W], p[W], y[8][W];
    if (VEC2) {
      const double2 tt = reinterpret_cast<const double2*>(trial)[i], pp = reinterpret_cast<const double2*>(prev)[i];
      t[0] = tt.x; t[W - 1] = tt.y; p[0] = pp.x; p[W - 1] = pp.y;
#pragma unroll
      for (int j = 0; j < 8; ++j) { y[j][0] = 0.0; y[j][W - 1] = 0.0; if (j < k.m && j != k.self) { const double2 yy = reinterpret_cast<const double2*>(k.y[j])[i]; y[j][0] = yy.x; y[j][W - 1] = yy.y; } }
    } else {
      t[0] = trial[i]; p[0] = prev[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) { y[j][0] = 0.0; if (j < k.m && j != k.self) y[j][0] = k.y[j][i]; }
    }
    double ei[W];
#pragma unroll
    for (int w = 0; w < W; ++w) ei[w] = t[w] - p[w];
    if (VEC2) {
      reinterpret_cast<double2*>(e)[i] = make_double2(ei[0], ei[W - 1]);
      if (xcopy) reinterpret_cast<double2*>(xcopy)[i] = make_double2(t[0], t[W - 1]);
    } else {
      e[i] = ei[0];
      if (xcopy) xcopy[i] = t[0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < k.m) {
#pragma unroll
      for (int w = 0; w < W; ++w) acc[j] += ei[w] * (j == k.self ? ei[w] : y[j][w]);
    }
  }
  for (int j = 0; j < k.m; ++j) {
    const double r = block_reduce<false>(acc[j], sh);
    if (threadIdx.x == 0) partial[(long long)j * np + BID.x] = r;
    __syncthreads();
  }
  if (!last_workgroup(counter, (unsigned)np, finish)) return;
  for (int j = 0; j < k.m; ++j) {
    double a = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) a += __builtin_nontemporal_load(partial + (long long)j * np + i);
    const double r = block_reduce<false>(a, sh);
    if (threadIdx.x == 0) { row_dev[j] = r; row_host[j] = r; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *counter = 0u; publish_flag(flag_host, seq); }
}
template <bool VEC2>
__global__ void __launch_bounds__(256) diis_push_kernel(long long n, const double* trial, const double* prev, double* e, double* xcopy, DiisPushK k, double* partial,
                                                        unsigned* counter, double* row_dev, double* row_host, int finish, unsigned long long* flag_host, unsigned long long seq) {
  diis_push_kernel_body<VEC2>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), n, trial, prev, e, xcopy, k, partial, counter, row_dev, row_host, finish, flag_host, seq);
}
int dev_diis_push(int64_t n, const double* trial, const double* prev, double* e, double* xcopy, int m, const double* const* ys, int self,
                  double* row_dev, double* row_host, void* flag_host, unsigned long long seq) {
  REQUIRE_INIT();
  if (m <= 0 || m > 8 || self < 0 || self >= m) { set_error("dev_diis_push: 1 <= m <= 8 vectors, 0 <= self < m"); return QEMB_ERR_ARG; }
  DiisPushK k{};
  k.m = m; k.self = self;
  bool vec2 = (n % 2 == 0);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  vec2 = vec2 && al16(trial) && al16(prev) && al16(e) && (!xcopy || al16(xcopy));
  for (int j = 0; j < m; ++j) { k.y[j] = ys[j]; vec2 = vec2 && al16(ys[j]); }
  const int64_t nw = vec2 ? n / 2 : n;
  const int np = (int)std::max<int64_t>(1, std::min<int64_t>((nw + 1023) / 1024, NPART));
  if (t_batch.on) {
    BatchEntry b{};
    b.kind = vec2 ? 1 : 0; b.gx = (unsigned)np; b.gy = 1; b.lds = 0;
    b.n = n; b.trial = trial; b.prev = prev; b.e = e; b.xcopy = xcopy; b.pk = k;
    b.out_dev = row_dev; b.out_host = row_host; b.finish = 0; b.flag = (unsigned long long*)flag_host; b.seq = seq; b.fin_m = m; b.fin_np = np;
    t_batch.entries.push_back(b);
    return QEMB_OK;
  }
  if (vec2) hipLaunchKernelGGL(diis_push_kernel<true>, dim3(np), dim3(256), 0, g_stream, (long long)n, trial, prev, e, xcopy, k, g_partials, (unsigned*)(g_partials + 8 * NPART), row_dev, row_host, post_finish_mode(),
                               (unsigned long long*)flag_host, seq);
  else hipLaunchKernelGGL(diis_push_kernel<false>, dim3(np), dim3(256), 0, g_stream, (long long)n, trial, prev, e, xcopy, k, g_partials, (unsigned*)(g_partials + 8 * NPART), row_dev, row_host, post_finish_mode(),
                          (unsigned long long*)flag_host, seq);
  if (post_finish_mode() == 0) hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(256), 0, g_stream, m, np, (const double*)g_partials, row_dev, row_host, (unsigned long long*)flag_host, seq);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// grid (o * o tiles (i,j), chunks of `rows` rows a): the workgroup forms the new t1[i, its rows] and t1[j, :] in LDS, then streams its rows of the tile --
// every load of a step before the stores (amp may be x[0]), 16-byte accesses when v is even
template <bool VEC2>
__device__ __forceinline__ void ccsd_extrapolate_energy_kernel_body(const uint3 BID, const uint3 GDIM, int o, int v, int rows, ExtrapK k, double* amp, const double* __restrict__ L,
                                                                      double* __restrict__ tau, double* partial, unsigned* counter, double* e_dev, double* e_host, int finish,
                                                                      unsigned long long* flag_host, unsigned long long seq) {
  extern __shared__ double t1rows[];                 // the new t1[i, a0 .. a1) and t1[j, :]
  __shared__ double sh[4];
  constexpr int W = VEC2 ? 2 : 1;
  const int i = (int)BID.x / o, j = (int)BID.x - i * o;
  const int a0 = (int)BID.y * rows, a1 = min(v, a0 + rows);
  double* ti = t1rows;
  double* tj = t1rows + rows;
  for (int a = threadIdx.x; a < v + (a1 - a0); a += blockDim.x) {
    const bool is_j = a < v;
    const long long src = is_j ? (long long)j * v + a : (long long)i * v + a0 + (a - v);
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q < k.n) sacc += k.c[q] * k.x[q][src];
    if (is_j) { tj[a] = sacc; if (k.write_amp && i == 0 && BID.y == 0) amp[src] = sacc; }     // the (0, j) tiles' first chunks store the new t1[j, :]
    else ti[a - v] = sacc;
  }
  __syncthreads();
  const long long vv = (long long)v * v, off = (long long)BID.x * vv, nov = (long long)o * v;
  const int vw = v / W;
  double acc = 0.0;
  for (int t = threadIdx.x; t < (a1 - a0) * vw; t += blockDim.x) {
    const int ar = t / vw, bw = t - ar * vw;
    const long long pos = off + (long long)(a0 + ar) * v + (long long)bw * W;
    double xv[8][W], lv[W];
    if (VEC2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < k.n) { const double2 xx = *reinterpret_cast<const double2*>(k.x[q] + nov + pos); xv[q][0] = xx.x; xv[q][W - 1] = xx.y; }
      const double2 ll = *reinterpret_cast<const double2*>(L + pos); lv[0] = ll.x; lv[W - 1] = ll.y;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < k.n) xv[q][0] = k.x[q][nov + pos];
      lv[0] = L[pos];
    }
    double val[W], tv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      val[w] = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < k.n) val[w] += k.c[q] * xv[q][w];
      tv[w] = val[w] + ti[ar] * tj[bw * W + w];
      acc += lv[w] * tv[w];
    }
    if (VEC2) {
      if (k.write_amp) *reinterpret_cast<double2*>(amp + nov + pos) = make_double2(val[0], val[W - 1]);
      *reinterpret_cast<double2*>(tau + pos) = make_double2(tv[0], tv[W - 1]);
    } else {
      if (k.write_amp) amp[nov + pos] = val[0];
      tau[pos] = tv[0];
    }
  }
  const unsigned np = GDIM.x * GDIM.y, me = BID.y * GDIM.x + BID.x;
  const double r = block_reduce<false>(acc, sh);
  if (threadIdx.x == 0) partial[me] = r;
  if (!last_workgroup(counter, np, finish)) return;
  double a = 0.0;
  for (unsigned q = threadIdx.x; q < np; q += blockDim.x) a += __builtin_nontemporal_load(partial + q);
  const double e = block_reduce<false>(a, sh);
  if (threadIdx.x == 0) { e_dev[0] = e; e_host[0] = e; *counter = 0u; publish_flag(flag_host, seq); }
}
template <bool VEC2>
__global__ void __launch_bounds__(256) ccsd_extrapolate_energy_kernel(int o, int v, int rows, ExtrapK k, double* amp, const double* __restrict__ L, double* __restrict__ tau,
                                                                      double* partial, unsigned* counter, double* e_dev, double* e_host, int finish, unsigned long long* flag_host, unsigned long long seq) {
  ccsd_extrapolate_energy_kernel_body<VEC2>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, rows, k, amp, L, tau, partial, counter, e_dev, e_host, finish, flag_host, seq);
}
int dev_ccsd_extrapolate_energy(int64_t o, int64_t v, int nterms, const double* coef, const double* const* xs, double* amp, const double* L,
                                double* tau, double* e_dev, double* e_host, void* flag_host, unsigned long long seq) {
  REQUIRE_INIT();
  if (nterms <= 0 || nterms > 8 || o <= 0 || v <= 0 || o * o > 8 * NPART || v > 4096) { set_error("dev_ccsd_extrapolate_energy: 1 <= nterms <= 8, o^2 <= 16384, v <= 4096"); return QEMB_ERR_ARG; }
  ExtrapK k{};
  k.n = nterms;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool vec2 = (v % 2 == 0) && ((o * v) % 2 == 0) && al16(amp) && al16(L) && al16(tau);
  for (int q = 0; q < nterms; ++q) { k.x[q] = xs[q]; k.c[q] = coef[q]; vec2 = vec2 && al16(xs[q]); }
  k.write_amp = !(nterms == 1 && xs[0] == amp && coef[0] == 1.0);
  // (i,j) tiles of v x v, cut into row chunks of ~4000 elements while the partials fit: several thousand workgroups balance over the 256 CUs, and the
  // t1 rows every workgroup forms first (v + rows values) stay a few per cent of what it streams
  int64_t chunks = std::max<int64_t>(1, std::min<int64_t>({(v * v + 3999) / 4000, (int64_t)(8 * NPART) / (o * o), v}));
  const int rows = (int)((v + chunks - 1) / chunks);
  chunks = (v + rows - 1) / rows;
  const size_t lds = (size_t)(v + rows) * sizeof(double);
  if (t_batch.on) {
    BatchEntry b{};
    b.kind = vec2 ? 3 : 2; b.gx = (unsigned)(o * o); b.gy = (unsigned)chunks; b.lds = lds;
    b.o = (int)o; b.v = (int)v; b.rows = rows; b.ek = k; b.amp = amp; b.L = L; b.tau = tau;
    b.out_dev = e_dev; b.out_host = e_host; b.finish = 0; b.flag = (unsigned long long*)flag_host; b.seq = seq; b.fin_m = 1; b.fin_np = (int)(o * o * chunks);
    t_batch.entries.push_back(b);
    return QEMB_OK;
  }
  if (vec2) hipLaunchKernelGGL(ccsd_extrapolate_energy_kernel<true>, dim3((unsigned)(o * o), (unsigned)chunks), dim3(256), lds, g_stream, (int)o, (int)v, rows, k, amp, L, tau,
                               g_partials, (unsigned*)(g_partials + 8 * NPART), e_dev, e_host, post_finish_mode(), (unsigned long long*)flag_host, seq);
  else hipLaunchKernelGGL(ccsd_extrapolate_energy_kernel<false>, dim3((unsigned)(o * o), (unsigned)chunks), dim3(256), lds, g_stream, (int)o, (int)v, rows, k, amp, L, tau,
                          g_partials, (unsigned*)(g_partials + 8 * NPART), e_dev, e_host, post_finish_mode(), (unsigned long long*)flag_host, seq);
  if (post_finish_mode() == 0) hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(256), 0, g_stream, 1, (int)(o * o * chunks), (const double*)g_partials, e_dev, e_host, (unsigned long long*)flag_host, seq);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_batch_begin() {
  REQUIRE_INIT();
  if (g_capturing || t_batch.on) { set_error("dev_batch_begin: capturing, or a batch is already open"); return QEMB_ERR_DEVICE; }
  t_batch.on = true;
  t_batch.entries.clear();
  return QEMB_OK;
}
static void register_groupable_kernels();
int dev_batch_flush() {
  REQUIRE_INIT();
  t_batch.on = false;
  std::vector<BatchEntry> ent;
  ent.swap(t_batch.entries);
  if (ent.empty()) return QEMB_OK;
  static std::once_flag once;
  std::lock_guard<std::mutex> reg_lock(g_plan_mutex);      // the registry of groupable kernels is filled lazily, also by the first dev_tape_run of another thread
  std::call_once(once, [] {
    register_groupable<diis_push_kernel_body<false>, 256, long long, const double*, const double*, double*, double*, DiisPushK, double*, unsigned*, double*, double*, int, unsigned long long*, unsigned long long>((const void*)diis_push_kernel<false>);
    register_groupable<diis_push_kernel_body<true>, 256, long long, const double*, const double*, double*, double*, DiisPushK, double*, unsigned*, double*, double*, int, unsigned long long*, unsigned long long>((const void*)diis_push_kernel<true>);
    register_groupable<ccsd_extrapolate_energy_kernel_body<false>, 256, int, int, int, ExtrapK, double*, const double*, double*, double*, unsigned*, double*, double*, int, unsigned long long*, unsigned long long>((const void*)ccsd_extrapolate_energy_kernel<false>);
    register_groupable<ccsd_extrapolate_energy_kernel_body<true>, 256, int, int, int, ExtrapK, double*, const double*, double*, double*, unsigned*, double*, double*, int, unsigned long long*, unsigned long long>((const void*)ccsd_extrapolate_energy_kernel<true>);
    register_groupable<finish_partials_kernel_body, 256, int, int, const double*, double*, double*, unsigned long long*, unsigned long long>((const void*)finish_partials_kernel);
  });
  if (!batch_region(ent.size() - 1)) { set_error("dev_batch_flush: scratch allocation failed"); return QEMB_ERR_ALLOC; }
  const void* wrappers[4] = {(const void*)diis_push_kernel<false>, (const void*)diis_push_kernel<true>, (const void*)ccsd_extrapolate_energy_kernel<false>,
                             (const void*)ccsd_extrapolate_energy_kernel<true>};
  for (size_t k = 0; k < ent.size(); ++k) {
    BatchEntry& b = ent[k];
    b.partial = batch_region(k); b.counter = (unsigned*)(b.partial + 8 * NPART);
    if (b.kind < 2) {
      void* pp[13] = {&b.n, &b.trial, &b.prev, &b.e, &b.xcopy, &b.pk, &b.partial, &b.counter, &b.out_dev, &b.out_host, &b.finish, &b.flag, &b.seq};
      std::memcpy(b.params, pp, sizeof(pp));
    } else {
      void* pp[14] = {&b.o, &b.v, &b.rows, &b.ek, &b.amp, &b.L, &b.tau, &b.partial, &b.counter, &b.out_dev, &b.out_host, &b.finish, &b.flag, &b.seq};
      std::memcpy(b.params, pp, sizeof(pp));
    }
    void* fp[7] = {&b.fin_m, &b.fin_np, &b.partial, &b.out_dev, &b.out_host, &b.flag, &b.seq};
    std::memcpy(b.fparams, fp, sizeof(fp));
  }
  std::vector<unsigned char> argbuf;
  auto issue = [&](const void* func, const std::vector<BatchEntry*>& mem, bool finish_kernel) -> int {
    for (size_t a = 0; a < mem.size(); a += GROUP_MAX) {
      const size_t nb = std::min<size_t>(GROUP_MAX, mem.size() - a);
      auto it = groupable().find(func);
      if (nb >= 2 && it != groupable().end()) {
        std::vector<GroupMember> gm;
        size_t lds = 0;
        for (size_t q = 0; q < nb; ++q) {
          BatchEntry* b = mem[a + q];
          gm.push_back(finish_kernel ? GroupMember{b->fparams, 1u, 1u, 1u} : GroupMember{b->params, b->gx, b->gy, 1u});
          if (!finish_kernel) lds = std::max(lds, b->lds);
        }
        argbuf.resize(it->second.args_bytes + 16);
        unsigned char* dst = (unsigned char*)(((uintptr_t)argbuf.data() + 15) & ~(uintptr_t)15);
        const unsigned blocks = it->second.build(dst, gm.data(), (int)nb);
        it->second.launch(dst, blocks, dim3(256), lds, g_stream);
      } else {
        for (size_t q = 0; q < nb; ++q) {
          BatchEntry* b = mem[a + q];
          if (finish_kernel) HIP_TRY(hipLaunchKernel(func, dim3(1), dim3(256), b->fparams, 0, g_stream));
          else HIP_TRY(hipLaunchKernel(func, dim3(b->gx, b->gy), dim3(256), b->params, b->lds, g_stream));
        }
      }
    }
    return QEMB_OK;
  };
  for (int kind = 0; kind < 4; ++kind) {
    std::vector<BatchEntry*> mem;
    for (BatchEntry& b : ent) if (b.kind == kind) mem.push_back(&b);
    if (mem.empty()) continue;
    int rc = issue(wrappers[kind], mem, false);
    if (rc) return rc;
    rc = issue((const void*)finish_partials_kernel, mem, true);
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_absmax(int64_t n, const double* x, double* out_dev) {
  REQUIRE_INIT();
  const int np = (int)std::max<int64_t>(1, std::min<int64_t>((n + 1023) / 1024, NPART));
  hipLaunchKernelGGL(reduce_stage1<true>, dim3(np), dim3(256), 0, g_stream, (long long)n, x, x, g_partials);
  hipLaunchKernelGGL(reduce_stage2<true>, dim3(1), dim3(256), 0, g_stream, np, g_partials, out_dev);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// J/K style contractions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gemv_rows_kernel_body(const uint3 BID, const uint3 GDIM, long long rows, long long cols, const double* T, long long ldt,
                                                        const double* x, double* y, double alpha, double beta) {
  __shared__ double sh[4];
  for (long long r = BID.x; r < rows; r += GDIM.x) {
    const double* row = T + r * ldt;
    double acc = 0.0;
    for (long long c = threadIdx.x; c < cols; c += blockDim.x) acc += row[c] * x[c];
    const double s = block_reduce<false>(acc, sh);
    if (threadIdx.x == 0) y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s;
  }
}
__global__ void __launch_bounds__(256) gemv_rows_kernel(long long rows, long long cols, const double* T, long long ldt,
                                                        const double* x, double* y, double alpha, double beta) { gemv_rows_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), rows, cols, T, ldt, x, y, alpha, beta); }
int dev_gemv_rows(int64_t rows, int64_t cols, const double* T, int64_t ldt, const double* x, double* y, double alpha, double beta) {
  REQUIRE_INIT();
  if (rows <= 0) return QEMB_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(rows, 1 << 20);
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(grid), dim3(256), 0, g_stream, (long long)rows, (long long)cols, T, (long long)ldt, x, y, alpha, beta);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// two matrix-vector products into one result, one pass: y[r] = alpha (T1[r,:] . x1 + T2[r,:] . x2) + beta y[r]
__device__ __forceinline__ void gemv_rows2_kernel_body(const uint3 BID, const uint3 GDIM, long long rows, long long cols, const double* T1, long long ld1, const double* x1,
                                                         const double* T2, long long ld2, const double* x2, double* y, double alpha, double beta) {
  __shared__ double sh[4];
  for (long long r = BID.x; r < rows; r += GDIM.x) {
    const double* r1 = T1 + r * ld1;
    const double* r2 = T2 + r * ld2;
    double a1 = 0.0, a2 = 0.0;
    for (long long c = threadIdx.x; c < cols; c += blockDim.x) { a1 += r1[c] * x1[c]; a2 += r2[c] * x2[c]; }
    const double s = block_reduce<false>(a1 + a2, sh);
    if (threadIdx.x == 0) y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s;
  }
}
__global__ void __launch_bounds__(256) gemv_rows2_kernel(long long rows, long long cols, const double* T1, long long ld1, const double* x1, const double* T2, long long ld2, const double* x2,
                                                         double* y, double alpha, double beta) {
  gemv_rows2_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), rows, cols, T1, ld1, x1, T2, ld2, x2, y, alpha, beta);
}
int dev_gemv_rows2(int64_t rows, int64_t cols, const double* T1, int64_t ld1, const double* x1, const double* T2, int64_t ld2, const double* x2, double* y, double alpha, double beta) {
  REQUIRE_INIT();
  if (rows <= 0) return QEMB_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(rows, 1 << 20);
  hipLaunchKernelGGL(gemv_rows2_kernel, dim3(grid), dim3(256), 0, g_stream, (long long)rows, (long long)cols, T1, (long long)ld1, x1, T2, (long long)ld2, x2, y, alpha, beta);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// The four small products of the T1 equation in one launch.  grid (o, chunks of 64 rows a): every workgroup forms Q[i,:] for its i (o dot products over v),
// then one wave per row a:  t1n[i,a] = sum_c t1[i,c] Lvv[a,c] + sum_k (Q[i,k] - Loo[k,i]) t1[k,a]
__device__ __forceinline__ void ccsd_t1_small_kernel_body(const uint3 BID, const uint3 GDIM, int o, int v, const double* __restrict__ t1, const double* __restrict__ Lvv,
                                                            const double* __restrict__ Loo, const double* __restrict__ Fov, double* __restrict__ t1n) {
  __shared__ double w[1024];              // Q[i,k] - Loo[k,i], k < o <= 1024
  const int i = (int)BID.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const double* ti = t1 + (long long)i * v;
  for (int k = wave; k < o; k += nw) {
    double acc = 0.0;
    for (int c = lane; c < v; c += 64) acc += ti[c] * Fov[(long long)k * v + c];
    acc = wave_sum(acc);
    if (lane == 0) w[k] = acc - Loo[(long long)k * o + i];
  }
  __syncthreads();
  const int a1 = min(v, ((int)BID.y + 1) * 64);
  for (int a = (int)BID.y * 64 + wave; a < a1; a += nw) {
    double acc = 0.0;
    for (int c = lane; c < v; c += 64) acc += ti[c] * Lvv[(long long)a * v + c];
    for (int k = lane; k < o; k += 64) acc += w[k] * t1[(long long)k * v + a];
    acc = wave_sum(acc);
    if (lane == 0) t1n[(long long)i * v + a] = acc;
  }
}
__global__ void __launch_bounds__(256) ccsd_t1_small_kernel(int o, int v, const double* __restrict__ t1, const double* __restrict__ Lvv, const double* __restrict__ Loo,
                                                            const double* __restrict__ Fov, double* __restrict__ t1n) {
  ccsd_t1_small_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, t1, Lvv, Loo, Fov, t1n);
}
int dev_ccsd_t1_small(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, double* t1n) {
  REQUIRE_INIT();
  if (o <= 0 || v <= 0) return QEMB_OK;
  if (o > 1024 || (v + 63) / 64 > 65535) { set_error("dev_ccsd_t1_small: n_occ <= 1024"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(ccsd_t1_small_kernel, dim3((unsigned)o, (unsigned)((v + 63) / 64)), dim3(256), 0, g_stream, (int)o, (int)v, t1, Lvv, Loo, Fov, t1n);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// The right-hand side of the T1 equation, one workgroup per element r = (i,a): the o dot products Q[i,k] - Loo[k,i] first (waves over k), then every thread
// strides over the two long rows S[r,:], Lph1[r,:], the short rows of the small products and the slabs of the two long-K products; one block sum.
__device__ __forceinline__ void ccsd_t1_assemble_kernel_body(const uint3 BID, const uint3 GDIM, int o, int v, const double* __restrict__ t1, const double* __restrict__ Lvv,
                                                               const double* __restrict__ Loo, const double* __restrict__ Fov, const double* __restrict__ Sm,
                                                               const double* __restrict__ Lph1, const double* __restrict__ PA, int SA, long long strideA,
                                                               const double* __restrict__ PB, int SB, long long strideB, double* __restrict__ t1n) {
  __shared__ double w[1024];              // Q[i,k] - Loo[k,i], k < o <= 1024
  __shared__ double sh[4];
  const long long nov = (long long)o * v;
  for (long long r = BID.x; r < nov; r += GDIM.x) {
    const int i = (int)(r / v), a = (int)(r - (long long)i * v);
    const double* ti = t1 + (long long)i * v;
    // Q[i,k] = sum_c t1[i,c] Fov[k,c]: eight lanes per k, 32 values of k at a time (a wave per k walked n_occ / 4 dependent rounds of loads: 7 us of the
    // 23 us this pass takes on the n_occ ~ 21 fragments of octane BE2)
    for (int k = threadIdx.x >> 3; k < o; k += blockDim.x >> 3) {
      double acc = 0.0;
      for (int c = threadIdx.x & 7; c < v; c += 8) acc += ti[c] * Fov[(long long)k * v + c];
      acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
      if ((threadIdx.x & 7) == 0) w[k] = acc - Loo[(long long)k * o + i];
    }
    const double* rs = Sm + r * nov;
    const double* rl = Lph1 + r * nov;
    double a1 = 0.0, a2 = 0.0;
    for (long long c = threadIdx.x; c < nov; c += blockDim.x) { a1 += rs[c] * Fov[c]; a2 += rl[c] * t1[c]; }
    double acc = a1 + a2;
    for (int c = threadIdx.x; c < v; c += blockDim.x) acc += ti[c] * Lvv[(long long)a * v + c];
    for (int sl = threadIdx.x; sl < SA; sl += blockDim.x) acc += PA[sl * strideA + r];
    for (int sl = threadIdx.x; sl < SB; sl += blockDim.x) acc -= PB[sl * strideB + r];
    __syncthreads();                      // w[] complete
    for (int k = threadIdx.x; k < o; k += blockDim.x) acc += w[k] * t1[(long long)k * v + a];
    const double tot = block_reduce<false>(acc, sh);
    if (threadIdx.x == 0) t1n[r] = tot;
  }
}
__global__ void __launch_bounds__(256) ccsd_t1_assemble_kernel(int o, int v, const double* __restrict__ t1, const double* __restrict__ Lvv, const double* __restrict__ Loo,
                                                               const double* __restrict__ Fov, const double* __restrict__ Sm, const double* __restrict__ Lph1,
                                                               const double* __restrict__ PA, int SA, long long strideA, const double* __restrict__ PB, int SB, long long strideB,
                                                               double* __restrict__ t1n) {
  ccsd_t1_assemble_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), o, v, t1, Lvv, Loo, Fov, Sm, Lph1, PA, SA, strideA, PB, SB, strideB, t1n);
}
int dev_ccsd_t1_assemble(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, const double* S, const double* Lph1,
                         const double* PA, int SA, int64_t strideA, const double* PB, int SB, int64_t strideB, double* t1n) {
  REQUIRE_INIT();
  if (o <= 0 || v <= 0) return QEMB_OK;
  if (o > 1024) { set_error("dev_ccsd_t1_assemble: n_occ <= 1024"); return QEMB_ERR_ARG; }
  hipLaunchKernelGGL(ccsd_t1_assemble_kernel, dim3((unsigned)std::min<int64_t>(o * v, 1 << 20)), dim3(256), 0, g_stream, (int)o, (int)v, t1, Lvv, Loo, Fov, S, Lph1,
                     PA, std::max(SA, 0), (long long)strideA, PB, std::max(SB, 0), (long long)strideB, t1n);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// two independent matrix-vector passes, one launch: the first rows1 workgroups take the first product, the rest the second
__device__ __forceinline__ void gemv_rows_two_kernel_body(const uint3 BID, const uint3 GDIM, long long rows1, long long cols1, const double* T1, long long ld1, const double* x1, double* y1,
                                                            double a1, double b1, long long rows2, long long cols2, const double* T2, long long ld2, const double* x2, double* y2,
                                                            double a2, double b2) {
  __shared__ double sh[4];
  for (long long r = BID.x; r < rows1 + rows2; r += GDIM.x) {
    const bool first = r < rows1;
    const long long rr = first ? r : r - rows1, cols = first ? cols1 : cols2;
    const double* row = first ? T1 + rr * ld1 : T2 + rr * ld2;
    const double* x = first ? x1 : x2;
    double acc = 0.0;
    for (long long c = threadIdx.x; c < cols; c += blockDim.x) acc += row[c] * x[c];
    const double s = block_reduce<false>(acc, sh);
    if (threadIdx.x == 0) {
      double* y = first ? y1 + rr : y2 + rr;
      const double al = first ? a1 : a2, be = first ? b1 : b2;
      *y = (be != 0.0) ? al * s + be * (*y) : al * s;
    }
  }
}
__global__ void __launch_bounds__(256) gemv_rows_two_kernel(long long rows1, long long cols1, const double* T1, long long ld1, const double* x1, double* y1, double a1, double b1,
                                                            long long rows2, long long cols2, const double* T2, long long ld2, const double* x2, double* y2, double a2, double b2) {
  gemv_rows_two_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), rows1, cols1, T1, ld1, x1, y1, a1, b1, rows2, cols2, T2, ld2, x2, y2, a2, b2);
}
int dev_gemv_rows_two(int64_t rows1, int64_t cols1, const double* T1, int64_t ld1, const double* x1, double* y1, double a1, double b1,
                      int64_t rows2, int64_t cols2, const double* T2, int64_t ld2, const double* x2, double* y2, double a2, double b2) {
  REQUIRE_INIT();
  if (rows1 + rows2 <= 0) return QEMB_OK;
  hipLaunchKernelGGL(gemv_rows_two_kernel, dim3((unsigned)std::min<int64_t>(rows1 + rows2, 1 << 20)), dim3(256), 0, g_stream, (long long)rows1, (long long)cols1, T1, (long long)ld1, x1, y1, a1, b1,
                     (long long)rows2, (long long)cols2, T2, (long long)ld2, x2, y2, a2, b2);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) gemv_rows_batched_kernel(long long rows, long long cols, long long nbatch, const double* __restrict__ T,
                                                               long long ldt, long long strideT, const double* __restrict__ x, long long stridex,
                                                               double* __restrict__ y, double alpha, double beta) {
  __shared__ double sh[4];
  const long long tot = nbatch * cols;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    double acc = 0.0;
    for (long long t = threadIdx.x; t < tot; t += blockDim.x) {
      const long long b = t / cols, c = t - b * cols;
      acc += T[b * strideT + r * ldt + c] * x[b * stridex + c];
    }
    const double s = block_reduce<false>(acc, sh);
    if (threadIdx.x == 0) y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s;
  }
}
int dev_gemv_rows_batched(int64_t rows, int64_t cols, int64_t nbatch, const double* T, int64_t ldt, int64_t strideT, const double* x,
                          int64_t stridex, double* y, double alpha, double beta) {
  REQUIRE_INIT();
  if (rows <= 0) return QEMB_OK;
  hipLaunchKernelGGL(gemv_rows_batched_kernel, dim3((unsigned)std::min<int64_t>(rows, 1 << 20)), dim3(256), 0, g_stream, (long long)rows, (long long)cols,
                     (long long)nbatch, T, (long long)ldt, (long long)strideT, x, (long long)stridex, y, alpha, beta);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// partial[p][chunk][r] = sum_{m in chunk} x[m] * T[p][m][r];  threads run along r (contiguous), blockIdx.z tiles r
__device__ __forceinline__ void contract_mid_stage1_body(const uint3 BID, const uint3 GDIM, long long mid, long long inner, int nchunk, const double* __restrict__ T,
                                                           const double* __restrict__ x, double* __restrict__ partial) {
  const long long p = BID.y;
  const int ch = BID.x;
  const long long m_per = (mid + nchunk - 1) / nchunk;
  const long long m0 = ch * m_per, m1 = (m0 + m_per < mid) ? m0 + m_per : mid;
  for (long long r = (long long)BID.z * blockDim.x + threadIdx.x; r < inner; r += (long long)GDIM.z * blockDim.x) {
    double a0 = 0.0, a1 = 0.0;
    const double* base = T + (p * mid) * inner + r;
    long long m = m0;
    for (; m + 1 < m1; m += 2) { a0 += x[m] * base[m * inner]; a1 += x[m + 1] * base[(m + 1) * inner]; }
    if (m < m1) a0 += x[m] * base[m * inner];
    partial[(p * nchunk + ch) * inner + r] = a0 + a1;
  }
}
__global__ void __launch_bounds__(256) contract_mid_stage1(long long mid, long long inner, int nchunk, const double* __restrict__ T,
                                                           const double* __restrict__ x, double* __restrict__ partial) { contract_mid_stage1_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), mid, inner, nchunk, T, x, partial); }
__device__ __forceinline__ void contract_mid_stage2_body(const uint3 BID, const uint3 GDIM, long long outer, long long inner, int nchunk, const double* partial,
                                                           double* Y, long long ldy, double alpha, double beta) {
  const long long t = (long long)BID.x * blockDim.x + threadIdx.x;
  if (t >= outer * inner) return;
  const long long p = t / inner, r = t - p * inner;
  double acc = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) acc += partial[(p * nchunk + ch) * inner + r];
  double* y = Y + p * ldy + r;
  *y = (beta != 0.0) ? alpha * acc + beta * (*y) : alpha * acc;
}
__global__ void __launch_bounds__(256) contract_mid_stage2(long long outer, long long inner, int nchunk, const double* partial,
                                                           double* Y, long long ldy, double alpha, double beta) { contract_mid_stage2_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), outer, inner, nchunk, partial, Y, ldy, alpha, beta); }
int dev_contract_mid(int64_t outer, int64_t mid, int64_t inner, const double* T, const double* x, double* Y, int64_t ldy, double alpha, double beta) {
  REQUIRE_INIT();
  if (outer <= 0 || inner <= 0) return QEMB_OK;
  if (outer > 65535) { set_error("dev_contract_mid: outer too large"); return QEMB_ERR_ARG; }
  // enough (p, chunk) workgroups to cover the chip: ~2048 in total, at least 8 rows of T per chunk
  const int64_t rblocks = std::min<int64_t>((inner + 255) / 256, 1024);
  int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(mid / 8, 128), std::max<int64_t>(1, (4096 + outer * rblocks - 1) / (outer * rblocks))));   // <= 128 partial slabs: stage 2 sums them serially
  int rc = ensure_ws((size_t)outer * nchunk * inner * sizeof(double));
  if (rc) return rc;
  hipLaunchKernelGGL(contract_mid_stage1, dim3(nchunk, (unsigned)outer, (unsigned)rblocks), dim3(256), 0, g_stream, (long long)mid, (long long)inner, nchunk, T, x, g_ws);
  const long long tot = outer * inner;
  hipLaunchKernelGGL(contract_mid_stage2, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, g_stream, (long long)outer, (long long)inner, nchunk, g_ws, Y, (long long)ldy, alpha, beta);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// exchange matrix from pair rows: stage 1 = per-row partial vectors, stage 2 = fixed-order sum over the rows that feed K[p,:]
__global__ void __launch_bounds__(256) k_pairs_stage1(long long n, const double* __restrict__ H, const double* __restrict__ D,
                                                      double* __restrict__ P1, double* __restrict__ P2) {
  extern __shared__ double sd[];            // D[q,:], D[p,:]
  const long long pq = blockIdx.x;
  long long p, q; unpair_ge(pq, p, q);
  for (long long s = threadIdx.x; s < n; s += blockDim.x) { sd[s] = D[q * n + s]; sd[n + s] = D[p * n + s]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const double* X = H + pq * n * n;
  for (long long r = wave; r < n; r += nw) {
    const double* row = X + r * n;
    double a1 = 0.0, a2 = 0.0;
    for (long long s = lane; s < n; s += 64) { const double x = row[s]; a1 += x * sd[s]; a2 += x * sd[n + s]; }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { P1[pq * n + r] = a1; P2[pq * n + r] = a2; }
  }
}
__global__ void __launch_bounds__(256) k_pairs_stage2(long long n, const double* __restrict__ P1, const double* __restrict__ P2, double* __restrict__ K) {
  const long long p = blockIdx.x;
  for (long long r = threadIdx.x; r < n; r += blockDim.x) {
    double acc = 0.0;
    for (long long q = 0; q <= p; ++q) acc += P1[(p * (p + 1) / 2 + q) * n + r];
    for (long long q = p + 1; q < n; ++q) acc += P2[(q * (q + 1) / 2 + p) * n + r];
    K[p * n + r] = acc;
  }
}
int dev_k_from_pairs(int64_t n, const double* H, const double* D, double* K) {
  REQUIRE_INIT();
  if (n <= 0) return QEMB_OK;
  const long long np = n * (n + 1) / 2;
  int rc = ensure_ws((size_t)2 * np * n * sizeof(double));
  if (rc) return rc;
  double* P1 = g_ws; double* P2 = g_ws + np * n;
  hipLaunchKernelGGL(k_pairs_stage1, dim3((unsigned)np), dim3(256), (size_t)2 * n * sizeof(double), g_stream, (long long)n, H, D, P1, P2);
  hipLaunchKernelGGL(k_pairs_stage2, dim3((unsigned)n), dim3(256), 0, g_stream, (long long)n, (const double*)P1, (const double*)P2, K);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// Coulomb AND exchange matrix in ONE pass over the 4-fold packed block S4[P(p,q)][P(r,s)] (a quarter of the n^4 tensor, half of the
// pair-row tensor dev_k_from_pairs reads, and the Coulomb product no longer needs its own pass).  One workgroup per packed row (p,q): the
// row is the lower triangle of the symmetric slab M[r,s] = (pq|rs), streamed ONCE from memory straight into registers -- wave w takes the
// rows r = w, w + 4, ..., lanes run over s <= r (contiguous) -- and every element is used for both halves of the symmetric product
//   y[r] = sum_{s<=r} M[r,s] d[s]  (reduced over the lanes of the wave that owns row r)
//        + sum_{s>r} M[s,r] d[s]   (accumulated by the lane that owns column r, over the rows its wave walks; the four waves' column
//                                    sums are added in a fixed order at the end)
// for d = D[q,:] (feeds K[p,:]) and d = D[p,:] (feeds K[q,:]), and for Jp[pq] = sum_{r>=s} M[r,s] Dp[rs], Dp = D + D^T off the diagonal.
// No LDS in the main loop, no atomics, every sum in a fixed order (run-to-run bitwise identical, like dev_k_from_pairs).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take_f64(double x) {      // the value of another lane (DPP pattern CTRL); 0 in rows ROW_MASK leaves out
  const unsigned long long u = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// sum over the 64 lanes without the LDS crossbar (__shfl_down = ds_bpermute: 12 LDS operations per double): butterfly inside each row of
// 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then row_bcast:15 into rows 1 / 3 and row_bcast:31 into rows 2 / 3;
// lane 63 holds the total, read back as a wave-uniform value
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_take_f64<0xB1, 0xF>(v);
  v += dpp_take_f64<0x4E, 0xF>(v);
  v += dpp_take_f64<0x141, 0xF>(v);
  v += dpp_take_f64<0x140, 0xF>(v);
  v += dpp_take_f64<0x142, 0xA>(v);
  v += dpp_take_f64<0x143, 0xC>(v);
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int NK>      // NK = ceil(n / 64) column slots per lane
__global__ void __launch_bounds__(256) jk_packed_stage1(int n, const double* __restrict__ S4, const double* __restrict__ D,
                                                        const double* __restrict__ Dp, double* __restrict__ Jp, double* __restrict__ P1,
                                                        double* __restrict__ P2) {
  extern __shared__ double sm[];
  double* yl1 = sm; double* yl2 = sm + n; double* part = sm + 2 * n;      // part[wave][2][n]; jw behind it
  double* jw = part + 8 * n;
  const long long np = (long long)n * (n + 1) / 2, pq = blockIdx.x;
  long long p, q; unpair_ge(pq, p, q);
  const double* __restrict__ row = S4 + pq * np;
  const double* __restrict__ Dq = D + q * n;
  const double* __restrict__ Dpr = D + p * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double dq[NK], dp[NK], uq[NK], up[NK], ja = 0.0;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int s = lane + 64 * k;
    dq[k] = s < n ? Dq[s] : 0.0; dp[k] = s < n ? Dpr[s] : 0.0; uq[k] = 0.0; up[k] = 0.0;
  }
  for (int r = wave; r < n; r += 4) {
    const int rb = r * (r + 1) / 2;                 // < 2^31 for n <= 1024
    const double dq_r = Dq[r], dp_r = Dpr[r];       // wave-uniform
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (64 * k <= r) {                            // wave-uniform
        const int s = lane + 64 * k;
        if (s <= r) {
          const double x = row[rb + s];
          a1 += x * dq[k]; a2 += x * dp[k];
          if (Jp) ja += x * Dp[rb + s];
          if (s < r) { uq[k] += x * dq_r; up[k] += x * dp_r; }
        }
      }
    }
    a1 = wave_sum_dpp(a1); a2 = wave_sum_dpp(a2);
    if (lane == 0) { yl1[r] = a1; yl2[r] = a2; }
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int s = lane + 64 * k;
    if (s < n) { part[(wave * 2 + 0) * n + s] = uq[k]; part[(wave * 2 + 1) * n + s] = up[k]; }
  }
  ja = wave_sum_dpp(ja);
  if (lane == 0) jw[wave] = ja;
  __syncthreads();
  for (int s = tid; s < n; s += 256) {
    P1[pq * n + s] = yl1[s] + ((part[0 * n + s] + part[2 * n + s]) + (part[4 * n + s] + part[6 * n + s]));
    P2[pq * n + s] = yl2[s] + ((part[1 * n + s] + part[3 * n + s]) + (part[5 * n + s] + part[7 * n + s]));
  }
  if (Jp && tid == 0) Jp[pq] = (jw[0] + jw[1]) + (jw[2] + jw[3]);
}

// Round 3: the same pass with enough bytes in flight to stream from HBM (the guide: ~72 KB per CU).  The kernel above keeps 8 bytes per lane
// and one row per wave in flight and pays two 64-lane reductions per row: 3.05 TB/s.  Here a wave works on FOUR consecutive rows at a time,
// one per 16-lane group; a lane holds the 16-byte piece (two columns) 2c + 32k of its row for every k, all NK2 pieces requested before the
// first is used (with the matching pieces of the packed density for the Coulomb sum: up to 14 loads of 16 bytes per lane outstanding, 8 waves
// per CU), and the row sums are reduced inside the 16-lane group with four row-local DPP steps for all four rows at once.  Column sums stay
// in the lane that owns the column pair; the 16 (wave, group) partial vectors are added in a fixed order at the end.  Rows start at any
// 8-byte offset: the 16-byte loads are issued at 8-byte alignment (global_load_dwordx4 needs dword alignment only).
struct alignas(8) D2u { double x, y; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for_i(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_i<I + 1, N>(f); }
}
__device__ __forceinline__ double row16_sum_dpp(double v) {      // total over the 16 lanes of a DPP row, in every lane of the row
  v += dpp_take_f64<0xB1, 0xF>(v);
  v += dpp_take_f64<0x4E, 0xF>(v);
  v += dpp_take_f64<0x141, 0xF>(v);
  v += dpp_take_f64<0x140, 0xF>(v);
  return v;
}
// GL = lanes per row group (16: four rows per wave step, 32: two); NKS = ceil(n / (2 GL)) 16-byte column slots per lane
template <int GL, int NKS, bool WITH_J>
__global__ void __launch_bounds__(256, 2) jk_packed_rowgroups(int n, const double* __restrict__ S4, const double* __restrict__ D,
                                                           const double* __restrict__ Dp, double* __restrict__ Jp, double* __restrict__ P1,
                                                           double* __restrict__ P2) {
  constexpr int G = 64 / GL, NSLOT = 4 * G, BAND = 2 * GL;      // row groups per wave; (wave, group) partial vectors; rows per band
  extern __shared__ double sm[];
  double* yl1 = sm; double* yl2 = sm + n; double* part = sm + 2 * n;      // part[(wave * G + group) * 2 + vector][n]; jw behind it
  double* jw = part + 2 * NSLOT * n;
  const long long np = (long long)n * (n + 1) / 2, pq = blockIdx.x;
  long long p, q; unpair_ge(pq, p, q);
  const double* __restrict__ row = S4 + pq * np;
  const double* __restrict__ Dq = D + q * n;
  const double* __restrict__ Dpr = D + p * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane / GL, c = lane % GL;
  double dq[NKS][2], dp[NKS][2], uq[NKS][2], up[NKS][2], ja = 0.0;
#pragma unroll
  for (int k = 0; k < NKS; ++k) {
    const int s = 2 * c + BAND * k;
    dq[k][0] = s < n ? Dq[s] : 0.0; dq[k][1] = s + 1 < n ? Dq[s + 1] : 0.0;
    dp[k][0] = s < n ? Dpr[s] : 0.0; dp[k][1] = s + 1 < n ? Dpr[s + 1] : 0.0;
    uq[k][0] = uq[k][1] = up[k][0] = up[k][1] = 0.0;
  }
  // Band b = rows BAND b .. BAND (b+1) - 1 needs the slots k <= b only: one fully unrolled body per band (KM = b + 1 slots), so that every
  // load of a wave step is issued before the first use and nothing is fetched beyond the triangle.  All loads are unconditional: addresses
  // are clamped into the packed row (np - 2 at most: a 16-byte load never leaves it) and the values masked afterwards.
  const int last = (int)np - 2;
  auto band = [&](auto kmc, int b) {
    constexpr int KM = decltype(kmc)::value;
    constexpr int STEPS = BAND / G;                    // wave steps per band
    const int qend = (STEPS * (b + 1) < (n + G - 1) / G) ? STEPS * (b + 1) : (n + G - 1) / G;
    // step Q = rows G Q .. G Q + G - 1, one per lane group
    auto fetch = [&](int Q, D2u (&x)[KM], D2u (&w)[KM]) {
      const int r = G * Q + g, rb = r * (r + 1) / 2;   // < 2^31 for n <= 1024
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        const int at = rb + 2 * c + BAND * k, cl = at < last ? at : last;
        x[k] = *reinterpret_cast<const D2u*>(row + cl);
        if (WITH_J) w[k] = *reinterpret_cast<const D2u*>(Dp + cl);
      }
    };
    auto consume = [&](int Q, const D2u (&x)[KM], const D2u (&w)[KM]) {
      const int r = G * Q + g;
      const bool valid = r < n;
      const int rb = r * (r + 1) / 2;
      const int rr = valid ? r : 0;
      const double dq_r = valid ? Dq[rr] : 0.0, dp_r = valid ? Dpr[rr] : 0.0;
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        const int s = 2 * c + BAND * k, at = rb + s;
        const bool shifted = at > last;               // only the very last element of the packed row: it arrived in the .y half
        double xx = shifted ? x[k].y : x[k].x, xy = x[k].y;
        double wx = 0.0, wy = 0.0;
        if (WITH_J) { wx = shifted ? w[k].y : w[k].x; wy = w[k].y; }
        xx = (valid && s <= r) ? xx : 0.0; xy = (valid && s + 1 <= r) ? xy : 0.0;
        a1 += xx * dq[k][0]; a1 += xy * dq[k][1];
        a2 += xx * dp[k][0]; a2 += xy * dp[k][1];
        if (WITH_J) { ja += xx * wx; ja += xy * wy; }
        const double cx = (s < r) ? xx : 0.0, cy = (s + 1 < r) ? xy : 0.0;      // the diagonal element belongs to the row sum only
        uq[k][0] += cx * dq_r; uq[k][1] += cy * dq_r;
        up[k][0] += cx * dp_r; up[k][1] += cy * dp_r;
      }
      a1 = row16_sum_dpp(a1); a2 = row16_sum_dpp(a2);
      if (GL == 32) {       // row_bcast:15 into DPP rows 1 and 3: their lanes then hold the sum over the 32 lanes of the group
        a1 += dpp_take_f64<0x142, 0xA>(a1); a2 += dpp_take_f64<0x142, 0xA>(a2);
      }
      if (c == GL - 1 && valid) { yl1[r] = a1; yl2[r] = a2; }
    };
    // (requesting the pieces of the wave's next step before the current one is consumed -- two register sets -- was measured: slower,
    //  1.49 vs 1.32 ms at n = 220 with 32-lane groups, no gain with 16-lane groups; the pass is not latency bound)
    for (int Q = STEPS * b + wave; Q < qend; Q += 4) {
      D2u x[KM], w[KM];
      fetch(Q, x, w);
      consume(Q, x, w);
    }
  };
  static_for_i<0, NKS>([&](auto kc) { band(std::integral_constant<int, decltype(kc)::value + 1>{}, decltype(kc)::value); });
  const int slot = wave * G + g;
#pragma unroll
  for (int k = 0; k < NKS; ++k) {
    const int s = 2 * c + BAND * k;
    if (s < n) { part[(slot * 2 + 0) * n + s] = uq[k][0]; part[(slot * 2 + 1) * n + s] = up[k][0]; }
    if (s + 1 < n) { part[(slot * 2 + 0) * n + s + 1] = uq[k][1]; part[(slot * 2 + 1) * n + s + 1] = up[k][1]; }
  }
  if (WITH_J) {
    ja = wave_sum_dpp(ja);
    if (lane == 0) jw[wave] = ja;
  }
  __syncthreads();
  for (int s = tid; s < n; s += 256) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl += 4) {            // fixed order
      t1 += (part[((sl + 0) * 2) * n + s] + part[((sl + 1) * 2) * n + s]) + (part[((sl + 2) * 2) * n + s] + part[((sl + 3) * 2) * n + s]);
      t2 += (part[((sl + 0) * 2 + 1) * n + s] + part[((sl + 1) * 2 + 1) * n + s]) + (part[((sl + 2) * 2 + 1) * n + s] + part[((sl + 3) * 2 + 1) * n + s]);
    }
    P1[pq * n + s] = yl1[s] + t1;
    P2[pq * n + s] = yl2[s] + t2;
  }
  if (WITH_J && tid == 0) Jp[pq] = (jw[0] + jw[1]) + (jw[2] + jw[3]);
}

int dev_jk_from_packed(int64_t n, const double* S4, const double* D, const double* Dp, double* Jp, double* K) {
  REQUIRE_INIT();
  if (n <= 0) return QEMB_OK;
  if (n > 1024) { set_error("dev_jk_from_packed: n > 1024 (use dev_k_from_pairs)"); return QEMB_ERR_ARG; }
  if (!S4 || !D || (Jp && !Dp) || (!Jp && !K)) { set_error("dev_jk_from_packed: bad arguments"); return QEMB_ERR_ARG; }
  const long long np = n * (n + 1) / 2;
  int rc = ensure_ws((size_t)2 * np * n * sizeof(double));
  if (rc) return rc;
  double* P1 = g_ws; double* P2 = g_ws + np * n;
  hipError_t attr_err = hipSuccess;
  static const bool old_kernel = std::getenv("QEMB_JK_V1") != nullptr;      // A/B measurements (tools/hbm_kernels.py)
  static const int gl = std::getenv("QEMB_JK_GL") ? std::atoi(std::getenv("QEMB_JK_GL")) : 16;      // 16: 1.28 ms, 32: 1.32 ms at n = 220
  if (n >= 2 && n <= 448 && !old_kernel) {
    // 32-lane row groups (two rows per wave step): 8 partial vectors; 16-lane groups (four rows): 16
    const int nslot = gl == 16 ? 16 : 8;
    const size_t lds = (size_t)((2 + 2 * nslot) * n + 8) * sizeof(double);   // n = 220: 31.7 KB (GL 32) / 59.9 KB (GL 16) per workgroup
    auto launch = [&](auto kern) {
      if (lds > 64 * 1024) attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (attr_err == hipSuccess) hipLaunchKernelGGL(kern, dim3((unsigned)np), dim3(256), lds, g_stream, (int)n, S4, D, Dp, Jp, P1, P2);
    };
#define QEMB_JK_CASE(GL, NKS) do { if (Jp) launch(jk_packed_rowgroups<GL, NKS, true>); else launch(jk_packed_rowgroups<GL, NKS, false>); } while (0)
    if (gl == 16) {
      const int nks = (int)((n + 31) / 32);
      if (nks <= 1) QEMB_JK_CASE(16, 1);
      else if (nks <= 2) QEMB_JK_CASE(16, 2);
      else if (nks <= 4) QEMB_JK_CASE(16, 4);
      else if (nks <= 7) QEMB_JK_CASE(16, 7);
      else if (nks <= 10) QEMB_JK_CASE(16, 10);
      else QEMB_JK_CASE(16, 14);
    } else {
      const int nks = (int)((n + 63) / 64);
      if (nks <= 1) QEMB_JK_CASE(32, 1);
      else if (nks <= 2) QEMB_JK_CASE(32, 2);
      else if (nks <= 3) QEMB_JK_CASE(32, 3);
      else if (nks <= 4) QEMB_JK_CASE(32, 4);
      else if (nks <= 5) QEMB_JK_CASE(32, 5);
      else QEMB_JK_CASE(32, 7);
    }
#undef QEMB_JK_CASE
  } else {
    const size_t lds = (size_t)(10 * n + 8) * sizeof(double);
    const int nk = (int)((n + 63) / 64);
    auto launch = [&](auto kern) {
      // n > 818 would need more than the 64 KB of dynamic LDS a kernel gets by default (the packed block of such a fragment, > 0.9 TB, does not fit a device anyway)
      if (lds > 64 * 1024) attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (attr_err == hipSuccess) hipLaunchKernelGGL(kern, dim3((unsigned)np), dim3(256), lds, g_stream, (int)n, S4, D, Dp, Jp, P1, P2);
    };
    if (nk <= 1) launch(jk_packed_stage1<1>);
    else if (nk <= 2) launch(jk_packed_stage1<2>);
    else if (nk <= 4) launch(jk_packed_stage1<4>);
    else if (nk <= 8) launch(jk_packed_stage1<8>);
    else launch(jk_packed_stage1<16>);
  }
  HIP_TRY(attr_err);
  if (K) hipLaunchKernelGGL(k_pairs_stage2, dim3((unsigned)n), dim3(256), 0, g_stream, (long long)n, (const double*)P1, (const double*)P2, K);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// ------------------------------------------------------------------------------------------------
// packed-pair transforms.  pair(i,j) = i(i+1)/2 + j, i >= j  (reference shared/helper.py:260-276)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long pair_idx(long long i, long long j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// row walkers per tile of the tiled unpack: enough blocks for 8 per CU and some slack for the tail (QEMB_UNPACK_WALKERS overrides: A/B runs)
static int64_t unpack_walkers(int64_t rows, int64_t ntiles) {
  static const int64_t forced = std::getenv("QEMB_UNPACK_WALKERS") ? std::atoll(std::getenv("QEMB_UNPACK_WALKERS")) : 0;
  const int64_t want = forced > 0 ? forced : 65535;      // one row per block measured best (profiles/r03_hbm_kernels.jsonl)
  return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(rows, 65535), want));
}
// Tiled triangular unpack: one workgroup per packed row, 32 x 32 tiles of the lower triangle staged through LDS so that
// both the [k][l] image and its mirror [l][k] are written in 256-byte runs and every packed element is read once.
// dup == 1: the row index is itself a pair (p >= q) of an s4 block; the n x n image goes to rows (p,q) and (q,p).
// dup == 2: the SOURCE rows are gathered: packed row r = pair (x,y), x >= y, is read from row x*n + y (pair-row selection fused in).
__device__ __forceinline__ void unpack_tril_tiled_kernel_body(const uint3 BID, const uint3 GDIM, long long rows, long long n, const double* __restrict__ packed,
                                                               double* __restrict__ full, int dup, long long nr, long long ld) {
  // ld: row stride of the n x n images (>= n).  A stride that is a multiple of 16 doubles makes every 256-byte run of a tile start on a
  // 128-byte line: n = 220 unpacks at 3.7 TB/s with ld = 220 (every run ends in two partially written lines) and at 5.7 with ld = 224.
  __shared__ double tile[32][33];
  const long long np = n * (n + 1) / 2, n2 = n * ld;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // LB.x = lower-triangle tile (tk >= tl), LB.y (looped) = packed row: every block moves one 32 x 32 tile
  const uint3 LB = xcd_logical_block(BID, GDIM);
  long long tt = LB.x, tk, tl; unpair_ge(tt, tk, tl);
  for (long long r = LB.y; r < rows; r += GDIM.y) {
    const double* src = packed + r * np;
    double* dst0 = full + r * n2;
    double* dst1 = nullptr;
    if (dup == 1) {
      long long p, q; unpair_ge(r, p, q);
      dst0 = full + (p * n + q) * n2;
      if (p != q) dst1 = full + (q * n + p) * n2;
    } else if (dup == 2) {           // source rows are the x >= y rows of an (n*n)-row matrix: row r = pair (x,y) lives at x*n + y
      long long x, y; unpair_ge(r, x, y);
      src = packed + (x * nr + y) * np;
    }
    // the straight image is written from the registers the tile arrived in (before the barrier: it does not need LDS);
    // only the mirror image is read back transposed
    double xr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = ty + 8 * i;
      const long long k = tk * 32 + kk, l = tl * 32 + tx;
      xr[i] = (k < n && l <= k) ? src[k * (k + 1) / 2 + l] : 0.0;
      tile[kk][tx] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = ty + 8 * i;
      const long long k = tk * 32 + kk, l = tl * 32 + tx;
      if (k < n && l <= k) { dst0[k * ld + l] = xr[i]; if (dst1) dst1[k * ld + l] = xr[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = ty + 8 * i;
      const long long lr = tl * 32 + kk, kc = tk * 32 + tx;
      if (kc < n && lr < kc) { const double y = tile[tx][kk]; dst0[lr * ld + kc] = y; if (dst1) dst1[lr * ld + kc] = y; }
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) unpack_tril_tiled_kernel(long long rows, long long n, const double* __restrict__ packed,
                                                               double* __restrict__ full, int dup, long long nr, long long ld) { unpack_tril_tiled_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), rows, n, packed, full, dup, nr, ld); }

// s1[i,j,k,l] = s4[pair(i,j), pair(k,l)];  one block row per (i,j), threads along (k,l)
__global__ void __launch_bounds__(256) unpack_s4_kernel(long long n, const double* s4, double* s1) {
  const long long np = n * (n + 1) / 2, n2 = n * n;
  for (long long ij = blockIdx.x; ij < n2; ij += gridDim.x) {
    const long long i = ij / n, j = ij - i * n;
    const double* src = s4 + pair_idx(i, j) * np;
    double* dst = s1 + ij * n2;
    for (long long kl = threadIdx.x; kl < n2; kl += blockDim.x) {
      const long long k = kl / n, l = kl - k * n;
      dst[kl] = src[pair_idx(k, l)];
    }
  }
}
int dev_unpack_s4(int64_t n, const double* s4, double* s1) {
  REQUIRE_INIT();
  if (n >= 32) {
    const int64_t np = n * (n + 1) / 2;
    const int64_t nt = (n + 31) / 32;
    hipLaunchKernelGGL(unpack_tril_tiled_kernel, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)unpack_walkers(np, nt * (nt + 1) / 2)), dim3(256), 0, g_stream, (long long)np, (long long)n, s4, s1, 1, (long long)n, (long long)n);
  } else {
    hipLaunchKernelGGL(unpack_s4_kernel, dim3((unsigned)std::min<int64_t>(n * n, 1 << 20)), dim3(256), 0, g_stream, (long long)n, s4, s1);
  }
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) pack_s4_kernel(long long n, const double* s1, double* s4) {
  const long long np = n * (n + 1) / 2, n2 = n * n;
  for (long long ij = blockIdx.x; ij < np; ij += gridDim.x) {
    // unravel ij -> (i,j), i >= j
    long long i = (long long)((sqrt(8.0 * (double)ij + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > ij) --i;
    while ((i + 1) * (i + 2) / 2 <= ij) ++i;
    const long long j = ij - i * (i + 1) / 2;
    const double* src = s1 + (i * n + j) * n2;
    double* dst = s4 + ij * np;
    for (long long kl = threadIdx.x; kl < np; kl += blockDim.x) {
      long long k = (long long)((sqrt(8.0 * (double)kl + 1.0) - 1.0) * 0.5);
      while (k * (k + 1) / 2 > kl) --k;
      while ((k + 1) * (k + 2) / 2 <= kl) ++k;
      const long long l = kl - k * (k + 1) / 2;
      dst[kl] = src[k * n + l];
    }
  }
}
int dev_pack_s4(int64_t n, const double* s1, double* s4) {
  REQUIRE_INIT();
  const int64_t np = n * (n + 1) / 2;
  hipLaunchKernelGGL(pack_s4_kernel, dim3((unsigned)std::min<int64_t>(np, 1 << 20)), dim3(256), 0, g_stream, (long long)n, s1, s4);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) unpack_s8_kernel(long long np, const double* s8, double* s4) {
  for (long long r = blockIdx.x; r < np; r += gridDim.x)
    for (long long c = threadIdx.x; c < np; c += blockDim.x) s4[r * np + c] = s8[pair_idx(r, c)];
}
int dev_unpack_s8_to_s4(int64_t n, const double* s8, double* s4) {
  REQUIRE_INIT();
  const int64_t np = n * (n + 1) / 2;
  hipLaunchKernelGGL(unpack_s8_kernel, dim3((unsigned)std::min<int64_t>(np, 1 << 20)), dim3(256), 0, g_stream, (long long)np, s8, s4);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__device__ __forceinline__ void unpack_tril_rows_kernel_body(const uint3 BID, const uint3 GDIM, long long rows, long long n, const double* packed, double* full) {
  const long long np = n * (n + 1) / 2, n2 = n * n;
  for (long long r = BID.x; r < rows; r += GDIM.x)
    for (long long kl = threadIdx.x; kl < n2; kl += blockDim.x) {
      const long long k = kl / n, l = kl - k * n;
      full[r * n2 + kl] = packed[r * np + pair_idx(k, l)];
    }
}
__global__ void __launch_bounds__(256) unpack_tril_rows_kernel(long long rows, long long n, const double* packed, double* full) { unpack_tril_rows_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), rows, n, packed, full); }
int dev_unpack_tril_rows(int64_t rows, int64_t n, const double* packed, double* full) { return dev_unpack_tril_rows_ld(rows, n, n, packed, full); }
int dev_unpack_tril_rows_ld(int64_t rows, int64_t n, int64_t ld, const double* packed, double* full) {
  REQUIRE_INIT();
  if (rows <= 0) return QEMB_OK;
  if (ld < n) { set_error("dev_unpack_tril_rows_ld: ld < n"); return QEMB_ERR_ARG; }
  if (n >= 32 || ld != n)
  {
    const int64_t nt = (n + 31) / 32;
    hipLaunchKernelGGL(unpack_tril_tiled_kernel, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)unpack_walkers(rows, nt * (nt + 1) / 2)), dim3(256), 0, g_stream, (long long)rows, (long long)n, packed, full, 0, (long long)n, (long long)ld);
  }
  else
    hipLaunchKernelGGL(unpack_tril_rows_kernel, dim3((unsigned)std::min<int64_t>(rows, 1 << 20)), dim3(256), 0, g_stream, (long long)rows, (long long)n, packed, full);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
// full[P(x,y)][k][l] = in[(x*nr + y)][P(k,l)] for x >= y (x, y < nr; k, l < n): "keep the x >= y rows" and "unpack the pair column"
// in one pass
int dev_unpack_tril_pair_rows(int64_t nr, int64_t n, const double* in, double* full) { return dev_unpack_tril_pair_rows_ld(nr, n, n, in, full); }
int dev_unpack_tril_pair_rows_ld(int64_t nr, int64_t n, int64_t ld, const double* in, double* full) {
  REQUIRE_INIT();
  const int64_t np = n * (n + 1) / 2, npr = nr * (nr + 1) / 2;
  if (ld < n) { set_error("dev_unpack_tril_pair_rows_ld: ld < n"); return QEMB_ERR_ARG; }
  if (n < 32 && ld == n) {   // small problems: two simple passes through a staging buffer
    void* tmp = nullptr;
    int rc = dev_alloc(&tmp, sizeof(double) * (size_t)npr * np);
    if (rc) return rc;
    rc = dev_pack_pair_rows(nr, np, in, (double*)tmp);
    if (!rc) rc = dev_unpack_tril_rows(npr, n, (const double*)tmp, full);
    (void)dev_free(tmp);
    return rc;
  }
  const int64_t nt = (n + 31) / 32;
  hipLaunchKernelGGL(unpack_tril_tiled_kernel, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)unpack_walkers(npr, nt * (nt + 1) / 2)), dim3(256), 0, g_stream, (long long)npr, (long long)n, in, full, 2, (long long)nr, (long long)ld);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
__global__ void __launch_bounds__(256) pack_tril_rows_kernel(long long rows, long long n, const double* full, double* packed) {
  const long long np = n * (n + 1) / 2, n2 = n * n;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x)
    for (long long kl = threadIdx.x; kl < n2; kl += blockDim.x) {
      const long long k = kl / n, l = kl - k * n;
      if (k >= l) packed[r * np + k * (k + 1) / 2 + l] = full[r * n2 + kl];
    }
}
int dev_pack_tril_rows(int64_t rows, int64_t n, const double* full, double* packed) {
  REQUIRE_INIT();
  if (rows <= 0) return QEMB_OK;
  hipLaunchKernelGGL(pack_tril_rows_kernel, dim3((unsigned)std::min<int64_t>(rows, 1 << 20)), dim3(256), 0, g_stream, (long long)rows, (long long)n, full, packed);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

// every kernel of this file that exists in body + wrapper form (see "grouped launches" at the top)
static void register_groupable_kernels() {
  register_groupable<copy4_linear_kernel_body, 256, Copy4K>((const void*)copy4_linear_kernel);
  register_groupable<copy4_transpose_kernel_body, 256, Copy4K, int>((const void*)copy4_transpose_kernel);
  register_groupable<outer4_kernel_body, 256, Outer4K>((const void*)outer4_kernel);
  register_groupable<div_denom_kernel_body, 256, double*, long long, long long, long long, long long, const double*, const double*, const double*, const double*>((const void*)div_denom_kernel);
  register_groupable<ccsd_ph_layouts_kernel_body, 256, long long, long long, const double*, const double*, double*, double*, double*, double*, double*, double*, int>((const void*)ccsd_ph_layouts_kernel);
  register_groupable<small_k_update_kernel_body, 256, long long, long long, long long, double, const double*, long long, const double*, long long, double*, long long, int>((const void*)small_k_update_kernel);
  register_groupable<small_k_update_mfma_kernel_body, 256, int, int, int, double, const double*, long long, const double*, long long, double*, long long>((const void*)small_k_update_mfma_kernel);
  register_groupable<ccsd_y_traces_kernel_body, 256, long long, long long, const double*, const double*, double*, const double*, int, long long, double>((const void*)ccsd_y_traces_kernel);
  register_groupable<lincomb_kernel_body, 256, long long, LincombK, double, double*>((const void*)lincomb_kernel);
  register_groupable<ladder_pack_tau_kernel_body, 256, long long, long long, const double*, double*, long long, double*, long long>((const void*)ladder_pack_tau_kernel);
  register_groupable<scatter_pm_rows_kernel_body, 256, long long, long long, const double*, const double*, double*, const double*, int, long long, int, long long>((const void*)scatter_pm_rows_kernel);
  register_groupable<ccsd_finish_t2_kernel_body, 256, long long, long long, double*, const double*, const double*, const double*, const double*>((const void*)ccsd_finish_t2_kernel);
  register_groupable<ladder_scatter_pm_kernel_body, 256, long long, long long, const double*, long long, const double*, long long, double*, const double*, const double*, int, int, long long, int, long long, long long, long long>((const void*)ladder_scatter_pm_kernel);
  register_groupable<pack_w_pm_kernel_body, 256, long long, const double*, double*, long long, double*, long long>((const void*)pack_w_pm_kernel);
  register_groupable<pack_w_pm_sum_kernel_body, 256, long long, const double*, const double*, const double*, double*, long long, double*, long long>((const void*)pack_w_pm_sum_kernel);
  register_groupable<foo_from_x_kernel_body, 256, long long, const double*, double*>((const void*)foo_from_x_kernel);
  register_groupable<gemv_rows2_kernel_body, 256, long long, long long, const double*, long long, const double*, const double*, long long, const double*, double*, double, double>((const void*)gemv_rows2_kernel);
  register_groupable<ccsd_t1_small_kernel_body, 256, int, int, const double*, const double*, const double*, const double*, double*>((const void*)ccsd_t1_small_kernel);
  register_groupable<ccsd_t1_assemble_kernel_body, 256, int, int, const double*, const double*, const double*, const double*, const double*, const double*, const double*, int, long long, const double*, int, long long, double*>((const void*)ccsd_t1_assemble_kernel);
  register_groupable<gemv_rows_two_kernel_body, 256, long long, long long, const double*, long long, const double*, double*, double, double, long long, long long, const double*, long long, const double*, double*, double, double>((const void*)gemv_rows_two_kernel);
  register_groupable<ccsd_finish_t2_rings_kernel_body, 256, long long, long long, double*, const double*, const double*, const double*, const double*, const double*, const double*, double*>((const void*)ccsd_finish_t2_rings_kernel);
  register_groupable<gemv_rows_kernel_body, 256, long long, long long, const double*, long long, const double*, double*, double, double>((const void*)gemv_rows_kernel);
  register_groupable<contract_mid_stage1_body, 256, long long, long long, int, const double*, const double*, double*>((const void*)contract_mid_stage1);
  register_groupable<contract_mid_stage2_body, 256, long long, long long, int, const double*, double*, long long, double, double>((const void*)contract_mid_stage2);
  register_groupable<unpack_tril_rows_kernel_body, 256, long long, long long, const double*, double*>((const void*)unpack_tril_rows_kernel);
  register_groupable<unpack_tril_tiled_kernel_body, 256, long long, long long, const double*, double*, int, long long, long long>((const void*)unpack_tril_tiled_kernel);
  register_groupable<fill_kernel_body, 1024, double*, long long, double>((const void*)fill_kernel);
}

}  // namespace qemb
