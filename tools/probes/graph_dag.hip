// graph_dag.hip -- does a hipGraph with independent branches run them side by side on this runtime?  (round 5: the lock-step CCSD iteration of small
// fragments is ~46 dependent launches of 5-15 us; its data-flow graph is ~15 levels deep.)
//   chain: W x D small kernels in one chain;  dag: D levels of W independent kernels, every node of a level depending on every node of the level before;
//   streams: the same DAG issued by hand on W streams with events between levels.
// hipcc --offload-arch=gfx950 -O3 tools/probes/graph_dag.hip -o tools/probes/graph_dag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(double* x, int n, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  for (int r = 0; r < reps; ++r) v = v * 1.0000001 + 1e-9;
  x[i] = v;
}
int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 3, D = argc > 2 ? atoi(argv[2]) : 16, blocks = argc > 3 ? atoi(argv[3]) : 600, reps = argc > 4 ? atoi(argv[4]) : 200;
  const int n = blocks * 256;
  std::vector<double*> buf(W);
  for (int w = 0; w < W; ++w) { CK(hipMalloc(&buf[w], sizeof(double) * n)); CK(hipMemset(buf[w], 0, sizeof(double) * n)); }
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // --- plain launches, one chain
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipStreamSynchronize(s));
    const double t0 = now();
    for (int d = 0; d < D; ++d) for (int w = 0; w < W; ++w) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, s, buf[w], n, reps);
    CK(hipStreamSynchronize(s));
    if (rep == 2) std::printf("launches in one stream : %8.1f us for %d kernels (%.2f us each)\n", now() - t0, W * D, (now() - t0) / (W * D));
  }
  // --- graphs
  for (int mode = 0; mode < 2; ++mode) {      // 0: chain, 1: levels
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> prev, cur;
    std::vector<void*> argstore;
    struct Args { double* x; int n; int reps; };
    std::vector<Args*> keep;
    for (int d = 0; d < D; ++d) {
      cur.clear();
      for (int w = 0; w < W; ++w) {
        Args* a = new Args{buf[w], n, reps}; keep.push_back(a);
        void** params = new void*[3]{&a->x, &a->n, &a->reps};
        hipKernelNodeParams kp{};
        kp.func = (void*)work; kp.gridDim = dim3(blocks); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = params; kp.extra = nullptr;
        hipGraphNode_t node;
        std::vector<hipGraphNode_t> deps;
        if (mode == 0) { if (!cur.empty()) deps.push_back(cur.back()); else if (!prev.empty()) deps.push_back(prev.back()); }
        else deps = prev;
        CK(hipGraphAddKernelNode(&node, g, deps.data(), deps.size(), &kp));
        cur.push_back(node);
      }
      prev = cur;
    }
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipStreamSynchronize(s));
      const double t0 = now();
      CK(hipGraphLaunch(ge, s));
      const double t1 = now();
      CK(hipStreamSynchronize(s));
      if (rep == 3) std::printf("graph %-6s            : %8.1f us (launch call %.1f us) for %d kernels, %d levels x %d wide\n", mode ? "levels" : "chain", now() - t0, t1 - t0, W * D, D, W);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // --- by hand: W streams, events between levels
  {
    std::vector<hipStream_t> st(W);
    for (int w = 0; w < W; ++w) CK(hipStreamCreateWithFlags(&st[w], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(W * (D + 1));
    for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    for (int rep = 0; rep < 3; ++rep) {
      for (int w = 0; w < W; ++w) CK(hipStreamSynchronize(st[w]));
      const double t0 = now();
      for (int d = 0; d < D; ++d) {
        for (int w = 0; w < W; ++w) {
          if (d > 0) for (int w2 = 0; w2 < W; ++w2) if (w2 != w) CK(hipStreamWaitEvent(st[w], ev[(d - 1) * W + w2], 0));
          hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, st[w], buf[w], n, reps);
          CK(hipEventRecord(ev[d * W + w], st[w]));
        }
      }
      const double t1 = now();
      for (int w = 0; w < W; ++w) CK(hipStreamSynchronize(st[w]));
      if (rep == 2) std::printf("streams + events       : %8.1f us (issue %.1f us)\n", now() - t0, t1 - t0);
    }
  }
  return 0;
}
