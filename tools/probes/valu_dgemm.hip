// valu_dgemm.hip -- prototype of an FP64 GEMM on the vector ALUs of gfx950 (v_fmac_f64 with a scalar-register operand):
//   C[m][n] = sum_k S[k][m] * V(n,k)         S: [K][lds] with m contiguous (the wave-uniform operand, fetched by SCALAR loads),
//                                             V: [N][ldv] k-contiguous (V_KC) or [K][ldv] n-contiguous, staged through LDS.
// Each wave owns RM rows x (64 RN) columns: per k step RM scalar values x RN LDS values per lane -> RM*RN v_fmac_f64 per lane.
// Standalone: checks a small case against the host, then times the shapes of the CCSD hot path.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_dgemm.hip -o tools/probes/valu_dgemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int RM, int RN, int WAVES, int BK, bool V_KC, int DBG = 0>
__global__ void __launch_bounds__(WAVES * 64) dgemm_valu(const double* __restrict__ S, long long lds_, const double* __restrict__ V, long long ldv,
                                                         double* __restrict__ C, long long ldc, long long strideC, int M, int N, int K, int kchunk) {
  constexpr int BN = 64 * RN, BM = WAVES * RM, T = WAVES * 64, LDB = BN + 4;
  constexpr int TOTAL = BN * BK / 2, NCH = (TOTAL + T - 1) / T;
  __shared__ __attribute__((aligned(16))) double Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const double* __restrict__ sp = S + m0 + wave * RM;
  double* __restrict__ Cz = C + (long long)blockIdx.z * strideC;

  double c[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) c[i][j] = 0.0;

  d2 st[NCH];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      d2 v = {0.0, 0.0};
      if (TOTAL % T != 0 && chunk >= TOTAL) { st[q] = v; continue; }
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        if (n0 + r < N && k0 + cc < kend) v = *reinterpret_cast<const d2*>(V + (long long)(n0 + r) * ldv + k0 + cc);
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        if (k0 + r < kend && n0 + cc < N) v = *reinterpret_cast<const d2*>(V + (long long)(k0 + r) * ldv + n0 + cc);
      }
      st[q] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      if (TOTAL % T != 0 && chunk >= TOTAL) continue;
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        Bs[buf][cc][r] = st[q][0];
        Bs[buf][cc + 1][r] = st[q][1];
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        *reinterpret_cast<d2*>(&Bs[buf][r][cc]) = st[q];
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  double a[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i) a[i] = sp[(long long)min(kbeg, kend - 1) * lds_ + i];
  load_tile(kbeg);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int kb = kbeg + kt * BK;
    if (kt + 1 < nk && !(DBG & 2)) load_tile(kb + BK);
    // scalar operand software-pipelined one k step ahead (the s_loads of step k+1 are in flight under the FMAs of step k)
#pragma unroll 2
    for (int k = 0; k < BK; ++k) {
      double b[RN];
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = Bs[cur][k][j * 64 + lane];
      const int kn = (DBG & 1) ? 0 : min(kb + k + 1, kend - 1);
      const double* __restrict__ arow = sp + (long long)kn * lds_;
      double an[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) an[i] = arow[i];
#pragma unroll
      for (int i = 0; i < RM; ++i) {
#pragma unroll
        for (int j = 0; j < RN; ++j) c[i][j] = __builtin_fma(a[i], b[j], c[i][j]);
      }
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = an[i];
    }
    if (kt + 1 < nk && !(DBG & 2)) store_tile(cur ^ 1);
    if (!(DBG & 4)) __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int m = m0 + wave * RM + i;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + j * 64 + lane;
        if (n < N) Cz[(long long)m * ldc + n] = c[i][j];
      }
    }
  }
}

template <int RM, int RN, int WAVES_M, int WAVES_N, int BK, bool V_KC, int DBG = 0>
__global__ void __launch_bounds__(WAVES_M * WAVES_N * 64) dgemm_valu3(const double* __restrict__ S, long long lds_, const double* __restrict__ V, long long ldv,
                                                         double* __restrict__ C, long long ldc, long long strideC, int M, int N, int K, int kchunk) {
  constexpr int WAVES = WAVES_M * WAVES_N, WNC = 64 * RN, BN = WAVES_N * WNC, BM = WAVES_M * RM, T = WAVES * 64, LDB = BN + 4;
  constexpr int TOTAL = BN * BK / 2, NCH = (TOTAL + T - 1) / T;
  __shared__ __attribute__((aligned(16))) double Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const double* __restrict__ sp = S + m0 + wm * RM;
  double* __restrict__ Cz = C + (long long)blockIdx.z * strideC;

  double c[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) c[i][j] = 0.0;

  d2 st[NCH];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      d2 v = {0.0, 0.0};
      if (TOTAL % T != 0 && chunk >= TOTAL) { st[q] = v; continue; }
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        if (n0 + r < N && k0 + cc < kend) v = *reinterpret_cast<const d2*>(V + (long long)(n0 + r) * ldv + k0 + cc);
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        if (k0 + r < kend && n0 + cc < N) v = *reinterpret_cast<const d2*>(V + (long long)(k0 + r) * ldv + n0 + cc);
      }
      st[q] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      if (TOTAL % T != 0 && chunk >= TOTAL) continue;
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        Bs[buf][cc][r] = st[q][0];
        Bs[buf][cc + 1][r] = st[q][1];
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        *reinterpret_cast<d2*>(&Bs[buf][r][cc]) = st[q];
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  double a[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i) a[i] = sp[(long long)min(kbeg, kend - 1) * lds_ + i];
  load_tile(kbeg);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int kb = kbeg + kt * BK;
    if (kt + 1 < nk && !(DBG & 2)) load_tile(kb + BK);
    // scalar operand software-pipelined one k step ahead (the s_loads of step k+1 are in flight under the FMAs of step k)
#pragma unroll 2
    for (int k = 0; k < BK; ++k) {
      double b[RN];
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = Bs[cur][k][wn * WNC + j * 64 + lane];
      const int kn = (DBG & 1) ? 0 : min(kb + k + 1, kend - 1);
      const double* __restrict__ arow = sp + (long long)kn * lds_;
      double an[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) an[i] = arow[i];
#pragma unroll
      for (int i = 0; i < RM; ++i) {
#pragma unroll
        for (int j = 0; j < RN; ++j) c[i][j] = __builtin_fma(a[i], b[j], c[i][j]);
      }
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = an[i];
    }
    if (kt + 1 < nk && !(DBG & 2)) store_tile(cur ^ 1);
    if (!(DBG & 4)) __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int m = m0 + wm * RM + i;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * WNC + j * 64 + lane;
        if (n < N) Cz[(long long)m * ldc + n] = c[i][j];
      }
    }
  }
}

typedef double v8d __attribute__((ext_vector_type(8)));

// v2: the scalar operand is loaded by hand-placed s_load_dwordx16 one k step ahead (the compiler sinks scalar loads next to their use,
// which exposes the full L2 latency at every step); RM = 16 only.
template <int RN, int WAVES, int BK, bool V_KC>
__global__ void __launch_bounds__(WAVES * 64) dgemm_valu2(const double* __restrict__ S, long long lds_, const double* __restrict__ V, long long ldv,
                                                          double* __restrict__ C, long long ldc, long long strideC, int M, int N, int K, int kchunk) {
  constexpr int RM = 16;
  constexpr int BN = 64 * RN, BM = WAVES * RM, T = WAVES * 64, LDB = BN + 4;
  constexpr int TOTAL = BN * BK / 2, NCH = (TOTAL + T - 1) / T;
  __shared__ __attribute__((aligned(16))) double Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const double* __restrict__ sp = S + m0 + wave * RM;
  double* __restrict__ Cz = C + (long long)blockIdx.z * strideC;

  double c[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) c[i][j] = 0.0;

  d2 st[NCH];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      d2 v = {0.0, 0.0};
      if (TOTAL % T != 0 && chunk >= TOTAL) { st[q] = v; continue; }
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        if (n0 + r < N && k0 + cc < kend) v = *reinterpret_cast<const d2*>(V + (long long)(n0 + r) * ldv + k0 + cc);
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        if (k0 + r < kend && n0 + cc < N) v = *reinterpret_cast<const d2*>(V + (long long)(k0 + r) * ldv + n0 + cc);
      }
      st[q] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int chunk = tid + q * T;
      if (TOTAL % T != 0 && chunk >= TOTAL) continue;
      if (V_KC) {
        const int r = chunk / (BK / 2), cc = (chunk % (BK / 2)) * 2;
        Bs[buf][cc][r] = st[q][0];
        Bs[buf][cc + 1][r] = st[q][1];
      } else {
        const int r = chunk / (BN / 2), cc = (chunk % (BN / 2)) * 2;
        *reinterpret_cast<d2*>(&Bs[buf][r][cc]) = st[q];
      }
    }
  };
#define SLOAD(lo, hi, kk)                                                                          \
  {                                                                                                \
    const double* p_ = sp + (long long)min((kk), kend - 1) * lds_;                                 \
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=s"(lo), "=s"(hi) : "s"(p_)); \
  }
#define SWAIT(lo, hi) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi));

  const int nk = (kend - kbeg + BK - 1) / BK;
  v8d alo, ahi, nlo, nhi;
  SLOAD(alo, ahi, kbeg);
  load_tile(kbeg);
  store_tile(0);
  SWAIT(alo, ahi);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int kb = kbeg + kt * BK;
    if (kt + 1 < nk) load_tile(kb + BK);
    double b[RN], bn[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) b[j] = Bs[cur][0][j * 64 + lane];
#pragma unroll 2
    for (int k = 0; k < BK; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      SLOAD(nlo, nhi, kb + k + 1);
      const int kr = min(k + 1, BK - 1);                  // the last step re-reads its own row: keeps the loop body uniform
#pragma unroll
      for (int j = 0; j < RN; ++j) bn[j] = Bs[cur][kr][j * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) c[i][j] = __builtin_fma(alo[i], b[j], c[i][j]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) c[8 + i][j] = __builtin_fma(ahi[i], b[j], c[8 + i][j]);
      __builtin_amdgcn_sched_barrier(0);
      SWAIT(nlo, nhi);
      alo = nlo; ahi = nhi;
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = bn[j];
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int m = m0 + wave * RM + i;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + j * 64 + lane;
        if (n < N) Cz[(long long)m * ldc + n] = c[i][j];
      }
    }
  }
}

static bool g_v2 = false;
template <int RM, int RN, int WAVES, int BK, bool V_KC>
static float launch(const double* S, long long lds_, const double* V, long long ldv, double* C, long long ldc, int M, int N, int K, int ksplit, int reps) {
  constexpr int BN = 64 * RN, BM = WAVES * RM;
  if (g_v2 && RM == 16) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, ksplit);
    const int kchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dgemm_valu2<RN, WAVES, BK, V_KC>), grid, dim3(WAVES * 64), 0, 0, S, lds_, V, ldv, C, ldc, (long long)M * ldc, M, N, K, kchunk);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL((dgemm_valu2<RN, WAVES, BK, V_KC>), grid, dim3(WAVES * 64), 0, 0, S, lds_, V, ldv, C, ldc, (long long)M * ldc, M, N, K, kchunk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, ksplit);
  const int kchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((dgemm_valu<RM, RN, WAVES, BK, V_KC>), grid, dim3(WAVES * 64), 0, 0, S, lds_, V, ldv, C, ldc, (long long)M * ldc, M, N, K, kchunk);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((dgemm_valu<RM, RN, WAVES, BK, V_KC>), grid, dim3(WAVES * 64), 0, 0, S, lds_, V, ldv, C, ldc, (long long)M * ldc, M, N, K, kchunk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int RM, int RN, int WM, int WN, int BK, bool V_KC>
static void bench3(const char* tag, int M, int N, int K) {
  constexpr int BN = WN * 64 * RN, BM = WM * RM;
  const int Mp = (M + BM - 1) / BM * BM;
  double *dS, *dV, *dC;
  const long long ldv = V_KC ? K : N;
  hipMalloc((void**)&dS, (size_t)K * Mp * 8); hipMalloc((void**)&dV, (size_t)N * K * 8); hipMalloc((void**)&dC, (size_t)M * N * 8);
  hipMemset(dS, 0, (size_t)K * Mp * 8); hipMemset(dV, 0, (size_t)N * K * 8);
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((dgemm_valu3<RM, RN, WM, WN, BK, V_KC>), grid, dim3(WM * WN * 64), 0, 0, dS, (long long)Mp, dV, ldv, dC, (long long)N, 0LL, M, N, K, K);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((dgemm_valu3<RM, RN, WM, WN, BK, V_KC>), grid, dim3(WM * WN * 64), 0, 0, dS, (long long)Mp, dV, ldv, dC, (long long)N, 0LL, M, N, K, K);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::printf("v3 %-26s RM=%2d RN=%d waves %dx%d BK=%2d kc=%d tile %3dx%4d  M=%5d N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s\n", tag, RM, RN, WM, WN, BK, (int)V_KC, BM, BN, M, N, K, ms,
              2.0 * M * N * K / ms / 1e9);
  hipFree(dS); hipFree(dV); hipFree(dC);
}

template <int DBG>
static void dbg_bench(const char* tag) {
  constexpr int RM = 16, RN = 4, WAVES = 8, BK = 16;
  const int M = 4000, N = 4000, K = 4000;
  double *dS, *dV, *dC;
  hipMalloc((void**)&dS, (size_t)K * M * 8); hipMalloc((void**)&dV, (size_t)N * K * 8); hipMalloc((void**)&dC, (size_t)M * N * 8);
  hipMemset(dS, 0, (size_t)K * M * 8); hipMemset(dV, 0, (size_t)N * K * 8);
  dim3 grid((N + 255) / 256, (M + 127) / 128, 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((dgemm_valu<RM, RN, WAVES, BK, true, DBG>), grid, dim3(512), 0, 0, dS, (long long)M, dV, (long long)K, dC, (long long)N, 0LL, M, N, K, K);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((dgemm_valu<RM, RN, WAVES, BK, true, DBG>), grid, dim3(512), 0, 0, dS, (long long)M, dV, (long long)K, dC, (long long)N, 0LL, M, N, K, K);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::printf("DBG %d %-50s %8.3f ms  %6.1f TFLOP/s\n", DBG, tag, ms, 2.0 * M * N * K / ms / 1e9);
  hipFree(dS); hipFree(dV); hipFree(dC);
}

template <int RM, int RN, int WAVES, int BK, bool V_KC>
static void check() {
  const int M = 45, N = 301, K = 70, Mp = (M + RM * WAVES - 1) / (RM * WAVES) * (RM * WAVES);
  std::vector<double> S((size_t)K * Mp, 0.0), V(V_KC ? (size_t)N * K : (size_t)K * (N + 1)), C((size_t)M * N, -1.0), R((size_t)M * N, 0.0);
  const long long ldv = V_KC ? K : N + 1;
  srand(1);
  for (int k = 0; k < K; ++k) for (int m = 0; m < M; ++m) S[(size_t)k * Mp + m] = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : V) x = rand() / (double)RAND_MAX - 0.5;
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    double s = 0; for (int k = 0; k < K; ++k) s += S[(size_t)k * Mp + m] * (V_KC ? V[(size_t)n * ldv + k] : V[(size_t)k * ldv + n]);
    R[(size_t)m * N + n] = s;
  }
  double *dS, *dV, *dC;
  hipMalloc((void**)&dS, S.size() * 8); hipMalloc((void**)&dV, V.size() * 8 + 64); hipMalloc((void**)&dC, C.size() * 8);
  hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice);
  launch<RM, RN, WAVES, BK, V_KC>(dS, Mp, dV, ldv, dC, N, M, N, K, 1, 1);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  double err = 0; for (size_t i = 0; i < C.size(); ++i) err = fmax(err, fabs(C[i] - R[i]));
  std::printf("check RM=%d RN=%d WAVES=%d BK=%d V_KC=%d: max err %.2e %s\n", RM, RN, WAVES, BK, (int)V_KC, err, err < 1e-12 ? "ok" : "FAIL");
  hipFree(dS); hipFree(dV); hipFree(dC);
}

template <int RM, int RN, int WAVES, int BK, bool V_KC>
static void bench(const char* tag, int M, int N, int K, int ksplit) {
  const int Mp = (M + RM * WAVES - 1) / (RM * WAVES) * (RM * WAVES);
  double *dS, *dV, *dC;
  const long long ldv = V_KC ? K : N;
  hipMalloc((void**)&dS, (size_t)K * Mp * 8); hipMalloc((void**)&dV, (size_t)N * K * 8); hipMalloc((void**)&dC, (size_t)M * N * 8 * ksplit);
  hipMemset(dS, 0, (size_t)K * Mp * 8); hipMemset(dV, 0, (size_t)N * K * 8);
  const float ms = launch<RM, RN, WAVES, BK, V_KC>(dS, Mp, dV, ldv, dC, N, M, N, K, ksplit, 5);
  std::printf("%-34s RM=%2d RN=%d W=%2d BK=%2d kc=%d ksplit=%2d  M=%5d N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s\n", tag, RM, RN, WAVES, BK, (int)V_KC, ksplit, M, N, K, ms,
              2.0 * M * N * K / ms / 1e9);
  hipFree(dS); hipFree(dV); hipFree(dC);
}

int main() {
  dbg_bench<0>("full kernel");
  dbg_bench<1>("scalar operand always row 0 (scalar-cache hits)");
  dbg_bench<2>("no tile reload (global loads + LDS stores skipped)");
  dbg_bench<3>("neither");
  dbg_bench<7>("neither, no barriers");
  dbg_bench<4>("full but no barriers (wrong results, timing only)");
  bench3<16, 4, 8, 1, 16, true>("ring", 4000, 4000, 4000);
  bench3<16, 4, 4, 2, 16, true>("ring", 4000, 4000, 4000);
  bench3<16, 4, 2, 4, 8, true>("ring", 4000, 4000, 4000);
  bench3<16, 4, 1, 8, 4, true>("ring", 4000, 4000, 4000);
  bench3<16, 4, 2, 4, 8, false>("ring", 4000, 4000, 4000);
  bench3<16, 4, 1, 8, 4, false>("ring", 4000, 4000, 4000);
  bench3<16, 4, 2, 2, 16, true>("ring", 4000, 4000, 4000);

  bench3<16, 4, 1, 4, 8, true>("ring", 4000, 4000, 4000);
  for (int v2 = 0; v2 < 0; ++v2) {
  g_v2 = v2 != 0;
  std::printf("---- kernel %s\n", g_v2 ? "v2 (hand-placed scalar loads, one step ahead)" : "v1 (compiler-scheduled)");
  check<16, 4, 8, 16, true>(); check<16, 4, 8, 16, false>(); check<16, 4, 7, 16, true>(); check<8, 4, 8, 16, true>();
  bench<16, 4, 8, 16, true>("ring (ov)^3, V k-contig", 4000, 4000, 4000, 1);
  bench<16, 4, 8, 16, false>("ring (ov)^3, V n-contig", 4000, 4000, 4000, 1);
  bench<16, 4, 8, 32, true>("ring (ov)^3, V k-contig, BK 32", 4000, 4000, 4000, 1);
  bench<16, 4, 4, 16, true>("ring (ov)^3, 4 waves", 4000, 4000, 4000, 1);
  bench<8, 4, 8, 16, true>("ring (ov)^3, RM 8", 4000, 4000, 4000, 1);
  if (!g_v2) {
    check<8, 8, 8, 16, true>(); check<8, 8, 8, 8, false>(); check<4, 8, 8, 16, true>();
    bench<8, 8, 8, 16, true>("ring, RM 8 RN 8 (64 x 512 tile)", 4000, 4000, 4000, 1);
    bench<8, 8, 8, 8, true>("ring, RM 8 RN 8 BK 8", 4000, 4000, 4000, 1);
    bench<8, 8, 8, 16, false>("ring, RM 8 RN 8, V n-contig", 4000, 4000, 4000, 1);
    bench<8, 8, 4, 16, true>("ring, RM 8 RN 8, 4 waves", 4000, 4000, 4000, 1);
    bench<4, 8, 8, 16, true>("ring, RM 4 RN 8 (32 x 512 tile)", 4000, 4000, 4000, 1);
    bench<4, 8, 16, 8, true>("ring, RM 4 RN 8, 16 waves BK 8", 4000, 4000, 4000, 1);
  }
  bench<16, 4, 7, 16, true>("ladder (+), 7 waves x 2", 210, 20100, 20100, 3);
  bench<16, 4, 7, 16, true>("ladder (+), 7 waves x 2", 210, 20100, 20100, 6);
  bench<16, 4, 6, 16, true>("ladder (-), 6 waves x 2", 190, 19900, 19900, 3);
  bench<16, 4, 6, 16, true>("ladder (-), 6 waves x 2", 190, 19900, 19900, 6);
  }
  return 0;
}
