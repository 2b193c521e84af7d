// hbm_streams.hip -- what the memory system gives a streaming kernel as a function of its read / write mix (gfx950, one MI355X):
//   R reads and W writes of 128 MB streams per pass, 16 bytes per lane and access, grid-stride, no reuse; plain and non-temporal accesses.
// The HBM-bound passes of the CCSD update are mixes of this kind (ccsd_ph_layouts: 1 read, 6 writes; ccsd_extrapolate_energy: 6 reads, 3 writes; ...):
// the figure of the same mix here is their roof, rather than the 8 TB/s of the data sheet.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/hbm_streams.hip -o tools/probes/hbm_streams && ./tools/probes/hbm_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));
struct Ptrs { const d2v* r[8]; d2v* w[8]; };
template <int R, int W, int NT = 0>
__global__ void __launch_bounds__(256) mix_kernel(Ptrs p, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    d2v acc = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < R; ++k) { const d2v x = NT ? __builtin_nontemporal_load(&p.r[k][i]) : p.r[k][i]; acc.x += x.x; acc.y += x.y; }
    if (W == 0) { if (acc.x == 1.2345e300) p.w[0][i] = acc; }      // (keeps the loads)
#pragma unroll
    for (int k = 0; k < W; ++k) { const d2v y = {acc.x + k, acc.y - k}; if (NT) __builtin_nontemporal_store(y, &p.w[k][i]); else p.w[k][i] = y; }
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
template <int R, int W, int NT = 0>
static int run(Ptrs p, long long n2, int blocks_per_cu) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * blocks_per_cu;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((mix_kernel<R, W, NT>), dim3(grid), dim3(256), 0, 0, p, n2);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(a, 0));
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((mix_kernel<R, W, NT>), dim3(grid), dim3(256), 0, 0, p, n2);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  const double bytes = (double)(R + W) * n2 * 16.0;
  std::printf("{\"reads\": %d, \"writes\": %d, \"nontemporal\": %d, \"blocks_per_cu\": %d, \"GB\": %.3f, \"ms\": %.4f, \"TBps\": %.2f, \"frac_of_8_TBps\": %.3f}\n", R, W, NT, blocks_per_cu, bytes / 1e9, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
  std::fflush(stdout);
  return 0;
}
int main() {
  const long long n2 = (128LL << 20) / 16;      // 128 MB per stream
  Ptrs p{};
  for (int k = 0; k < 8; ++k) { void* q; CK(hipMalloc(&q, n2 * 16)); CK(hipMemset(q, 0, n2 * 16)); p.r[k] = (const d2v*)q; }
  for (int k = 0; k < 8; ++k) { void* q; CK(hipMalloc(&q, n2 * 16)); CK(hipMemset(q, 0, n2 * 16)); p.w[k] = (d2v*)q; }
  for (int bpc : {8, 16}) {
    if (run<1, 0>(p, n2, bpc) || run<4, 0>(p, n2, bpc) || run<0, 1>(p, n2, bpc) || run<0, 4>(p, n2, bpc) || run<1, 1>(p, n2, bpc) || run<1, 6>(p, n2, bpc) || run<2, 1>(p, n2, bpc) ||
        run<5, 1>(p, n2, bpc) || run<6, 3>(p, n2, bpc) || run<7, 1>(p, n2, bpc) || run<3, 3>(p, n2, bpc) ||
        run<0, 4, 1>(p, n2, bpc) || run<1, 6, 1>(p, n2, bpc) || run<6, 3, 1>(p, n2, bpc) || run<4, 0, 1>(p, n2, bpc) || run<5, 1, 1>(p, n2, bpc)) return 1;
  }
  return 0;
}
