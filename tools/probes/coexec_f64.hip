// coexec_f64.hip -- probe: do v_fma_f64 (VALU) and v_mfma_f64_16x16x4_f64 (matrix pipe) execute concurrently on gfx950, and what does the
// chip sustain when both are fed?  NV = VALU FMA instructions issued per MFMA (a 64-lane v_fma_f64 = 128 flop; one MFMA = 2048 flop).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coexec_f64.hip -o tools/probes/coexec_f64 && ./tools/probes/coexec_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ void __launch_bounds__(256) probe(double* out, int iters, long long* cyc) {
  const long long c0 = clock64();                 // s_memtime: shader-clock ticks on gfx950
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 1e-3 * (threadIdx.x + j);
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x, x = 0.999999, y = 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NM) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = __builtin_fma(v[(i * NV + j) & 15], x, y);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int j = 0; j < 16; ++j) s += v[j];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - c0;
}

template <int NM, int NV>
static void run(const char* label, int bpc) {
  double* out; hipMalloc((void**)&out, 64);
  long long* cyc; hipMalloc((void**)&cyc, sizeof(long long) * 256 * bpc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * bpc;
  hipLaunchKernelGGL((probe<NM, NV>), dim3(grid), dim3(256), 0, 0, out, 64, cyc);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NM, NV>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * 4;
  const double fm = NM ? waves * 8.0 * iters * 2048.0 : 0.0, fv = waves * 8.0 * iters * NV * 128.0;
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
  double mean = 0; for (long long c : h) mean += (double)c; mean /= grid;
  std::printf("%-28s blocks/CU %d  %8.3f ms   MFMA %6.1f TF   VALU %6.1f TF   total %6.1f TF   block cycles / kernel time = %.2f GHz\n", label, bpc, ms, fm / ms / 1e9,
              fv / ms / 1e9, (fm + fv) / ms / 1e9, mean / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int bpc : {1, 2, 4, 8}) {
    run<1, 0>("MFMA only", bpc);
    run<0, 16>("VALU only (16 fma / slot)", bpc);
    run<1, 2>("MFMA + 2 v_fma each", bpc);
    run<1, 4>("MFMA + 4 v_fma each", bpc);
    run<1, 8>("MFMA + 8 v_fma each", bpc);
    run<1, 16>("MFMA + 16 v_fma each", bpc);
  }
  return 0;
}
