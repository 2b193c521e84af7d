// comm_shm.cpp -- TEST INFRASTRUCTURE: the dev_comm_* interface of csrc/dev_ops.h over a POSIX shared-memory segment, so that the
// multi-rank host logic (quemb_amd/comm.py rendezvous, be_func_parallel, bench.py's launcher) runs on the GPU-less build container with
// the scalar mock device layer.  The product library implements the same interface on RCCL (csrc/comm_rccl.hip); nothing here is linked
// into libqemb_hip.so.
//
// Protocol: the 128-byte id names a segment /dev/shm/qemb_hc_<pid>_<nonce> created by dev_comm_unique_id.  An all-reduce copies each
// rank's buffer into its slot, meets at a generation barrier, lets every rank combine the slots in rank order (so every rank gets the
// bit-identical result, as with RCCL), and meets again before the slots are reused.  Waiting is bounded (QEMB_COMM_TIMEOUT_S, 6 h unless set, as in the product):
// a lost rank turns into an error, not a hang.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include "dev_ops.h"

namespace qemb {
namespace {
constexpr int64_t SLOT_ELEMS = 1 << 14;       // doubles per rank and pass (longer buffers go in several passes)
constexpr int MAX_WORLD = 64;
struct ShmHeader {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  uint32_t world;
};
struct ShmComm {
  int rank = 0, world = 1;
  std::string name;
  ShmHeader* hdr = nullptr;
  double* slots = nullptr;
  size_t bytes = 0;
};
ShmComm* g_comm = nullptr;
bool g_broken = false;      // a wait timed out: like the product's aborted communicator, every later collective fails at once
std::mutex g_mutex;

double timeout_s() {
  const char* e = std::getenv("QEMB_COMM_TIMEOUT_S");            // the product's knob (csrc/comm_rccl.hip); the older mock-only name still works
  if (!e) e = std::getenv("QEMB_HC_COMM_TIMEOUT_S");
  const double v = e ? std::atof(e) : 21600.0;      // the product's default: far above any rank-to-rank skew of a sweep
  return v > 0 ? v : 1e18;                            // 0: no wall-clock bound
}
size_t segment_bytes(int world) { return 4096 + (size_t)world * SLOT_ELEMS * sizeof(double); }

int barrier(ShmComm& c) {
  const uint32_t gen = c.hdr->generation.load(std::memory_order_acquire);
  if (c.hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c.world) {
    c.hdr->arrived.store(0, std::memory_order_relaxed);
    c.hdr->generation.store(gen + 1, std::memory_order_release);
    return QEMB_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const double lim = timeout_s();
  int spins = 0;
  while (c.hdr->generation.load(std::memory_order_acquire) == gen) {
    if (++spins < 200) { std::this_thread::yield(); continue; }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
      set_error("hostcheck comm: rank " + std::to_string(c.rank) + " waited " + std::to_string((int)lim) + " s at a barrier (a rank is gone)");
      return QEMB_ERR_DEVICE;
    }
  }
  return QEMB_OK;
}

int map_segment(const std::string& name, int world, bool create, ShmComm& c) {
  const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
  if (fd < 0) { set_error("hostcheck comm: shm_open(" + name + ") failed: " + strerror(errno)); return QEMB_ERR_DEVICE; }
  const size_t bytes = segment_bytes(create ? MAX_WORLD : world);
  if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); set_error("hostcheck comm: ftruncate failed"); return QEMB_ERR_ALLOC; }
  void* p = mmap(nullptr, segment_bytes(world), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { set_error("hostcheck comm: mmap failed"); return QEMB_ERR_ALLOC; }
  c.hdr = (ShmHeader*)p; c.slots = (double*)((char*)p + 4096); c.bytes = segment_bytes(world); c.name = name;
  return QEMB_OK;
}
}  // namespace

int dev_comm_unique_id(void* id128) {
  if (!id128) { set_error("qemb_comm_unique_id: null buffer"); return QEMB_ERR_ARG; }
  static std::atomic<unsigned> counter{0};
  char name[COMM_ID_BYTES];
  memset(name, 0, sizeof name);
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(name, sizeof name, "/qemb_hc_%d_%llx_%u", (int)getpid(), (unsigned long long)now, counter.fetch_add(1));
  // the segment is created here (zero-filled by ftruncate: counters start at 0) and sized for any world up to MAX_WORLD
  ShmComm tmp;
  if (int rc = map_segment(name, MAX_WORLD, true, tmp)) return rc;
  munmap((void*)tmp.hdr, tmp.bytes);
  memcpy(id128, name, COMM_ID_BYTES);
  return QEMB_OK;
}

int dev_comm_init(int rank, int world, const void* id128) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || !id128) { set_error("qemb_comm_init: need 0 <= rank < world <= 64 and an id"); return QEMB_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_comm) { set_error("qemb_comm_init: a communicator already exists (one per process; qemb_comm_destroy first)"); return QEMB_ERR_ARG; }
  char name[COMM_ID_BYTES + 1];
  memcpy(name, id128, COMM_ID_BYTES); name[COMM_ID_BYTES] = 0;
  ShmComm* c = new ShmComm();
  c->rank = rank; c->world = world;
  if (int rc = map_segment(name, world, false, *c)) { delete c; return rc; }
  if (int rc = barrier(*c)) { munmap((void*)c->hdr, c->bytes); delete c; return rc; }     // collective, like ncclCommInitRank
  if (rank == 0) shm_unlink(c->name.c_str());     // every rank has mapped it: the name can go, the memory lives as long as the mappings
  g_comm = c;
  return QEMB_OK;
}

int dev_comm_info(int* rank, int* world) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (rank) *rank = g_comm ? g_comm->rank : 0;
  if (world) *world = g_comm ? g_comm->world : 1;
  return QEMB_OK;
}

int dev_comm_allreduce(double* buf, int64_t n, int op) {
  if (n < 0 || (n > 0 && !buf) || (op != COMM_SUM && op != COMM_MAX)) { set_error("qemb_comm_allreduce: bad arguments"); return QEMB_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_broken) { set_error("qemb_comm_allreduce: the communicator was aborted after a failed or timed-out collective"); return QEMB_ERR_DEVICE; }
  if (!g_comm) { set_error("qemb_comm_allreduce: no communicator (qemb_comm_init)"); return QEMB_ERR_DEVICE; }
  ShmComm& c = *g_comm;
  for (int64_t off = 0; off < n; off += SLOT_ELEMS) {
    const int64_t m = std::min<int64_t>(SLOT_ELEMS, n - off);
    memcpy(c.slots + (size_t)c.rank * SLOT_ELEMS, buf + off, (size_t)m * sizeof(double));
    if (int rc = barrier(c)) { g_broken = true; return rc; }
    for (int64_t i = 0; i < m; ++i) {
      double acc = c.slots[i];
      for (int r = 1; r < c.world; ++r) {
        const double x = c.slots[(size_t)r * SLOT_ELEMS + i];
        acc = (op == COMM_SUM) ? acc + x : (x > acc ? x : acc);
      }
      buf[off + i] = acc;
    }
    if (int rc = barrier(c)) { g_broken = true; return rc; }
  }
  return QEMB_OK;
}

int dev_comm_destroy() {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_broken = false;
  if (!g_comm) return QEMB_OK;
  ShmComm* c = g_comm;
  g_comm = nullptr;
  munmap((void*)c->hdr, c->bytes);
  delete c;
  return QEMB_OK;
}

}  // namespace qemb
