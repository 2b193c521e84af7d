// valu_dgemm_inner.hip -- probe: the inner loop a VALU (v_fma_f64) DGEMM would run on gfx950 -- per k step 16 wave-uniform A values from
// scalar loads (SGPR operands of the FMAs), 4 B values per lane from LDS (conflict-free ds_read_b64), 64 v_fma_f64 per lane -- without
// the global->LDS staging.  Upper bound of what such a kernel can sustain, next to the 57-59 TFLOP/s of the f64 MFMA GEMMs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_dgemm_inner.hip -o tools/probes/valu_dgemm_inner
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RM = 16, RN = 4, BK = 32, WN = 64 * RN;   // wave tile: RM rows x 256 columns

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) inner(const double* __restrict__ At /* [BK][RM * WAVES] */, const double* __restrict__ Bg /* [BK][WN] */,
                                                    double* __restrict__ out, int reps) {
  __shared__ double Bs[BK][WN];
  for (int i = threadIdx.x; i < BK * WN; i += WAVES * 64) Bs[i / WN][i % WN] = Bg[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double* __restrict__ a_row = At + wave * RM;     // wave-uniform
  double c[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) c[i][j] = 0.0;
  for (int r = 0; r < reps; ++r) {
#pragma unroll 4
    for (int k = 0; k < BK; ++k) {
      double b[RN];
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = Bs[k][j * 64 + lane];
#pragma unroll
      for (int i = 0; i < RM; ++i) {
        const double a = a_row[k * RM * WAVES + i];        // uniform address -> s_load
#pragma unroll
        for (int j = 0; j < RN; ++j) c[i][j] = __builtin_fma(a, b[j], c[i][j]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) s += c[i][j];
  out[(size_t)blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int WAVES>
static void run(int bpc) {
  double *At, *Bg, *out;
  hipMalloc((void**)&At, sizeof(double) * BK * RM * WAVES); hipMalloc((void**)&Bg, sizeof(double) * BK * WN);
  const int grid = 256 * bpc;
  hipMalloc((void**)&out, sizeof(double) * grid * WAVES * 64);
  hipMemset(At, 0, sizeof(double) * BK * RM * WAVES); hipMemset(Bg, 0, sizeof(double) * BK * WN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 4000;
  hipLaunchKernelGGL((inner<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, At, Bg, out, 8);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((inner<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, At, Bg, out, reps);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * grid * WAVES * 64.0 * reps * BK * RM * RN;
  std::printf("waves/block %d  blocks/CU %d  (%d waves/SIMD)  %8.3f ms  %6.1f TFLOP/s\n", WAVES, bpc, WAVES * bpc / 4, ms, flops / ms / 1e9);
  hipFree(At); hipFree(Bg); hipFree(out);
}

int main() {
  run<4>(1); run<4>(2); run<8>(1); run<8>(2); run<4>(3);
  return 0;
}
