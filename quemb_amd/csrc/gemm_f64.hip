// gemm_f64.hip -- hand-written FP64 GEMM for gfx950 (MI355X, CDNA4) on v_mfma_f64_16x16x4_f64.
//
// This one kernel family carries every O(N^5)/O(N^6) contraction of the hot path: the CCSD
// pp-ladder  t2new[ij,ab] += tau[ij,cd] * W[ab,cd]  (reference: molbe/solver.py:907 ->
// PySCF ccsd.update_amps/_add_vvvv), the ph rings, the embedding->MO quarter transforms
// (solver.py:900), the AO->embedding transforms (mbe.py:1038, eri_onthefly.py:133-143) and the
// DF fit/contract (eri_sparse_DF.cpp:611-621).
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//  * wave64; each wave owns a (16*WM) x (16*WN) output sub-tile = WM*WN MFMA accumulators of 4 f64.
//    f64 MFMA on gfx950 issues once per 64 cycles per SIMD (78.6 TF chip peak == the FP64 vector peak),
//    so the operand traffic per MFMA is tiny; what matters is never starving the matrix pipe:
//    register-prefetched global loads for tile t+1 are issued before the MFMAs of tile t (T14 split),
//    LDS is double buffered, one barrier per K-tile.
//  * LDS images are chosen per operand storage order so that BOTH the 16-byte staging writes and the
//    ds_read_b64 fragment reads are bank-conflict free:
//      K-contiguous operand  -> image [row][BK+2]   (row stride 2*(BK+2) dwords: 36 for BK=16;
//                               rows*36 mod 64 are 16 distinct multiples of 4, +{0,2} for the 2 k's
//                               of a 32-lane group)
//      M/N-contiguous operand-> image [k][BMN+pad], (BMN+pad) == 16 (mod 32) so that the two k rows
//                               read by a 32-lane group sit in opposite halves of the 64 banks.
//  * f64 MFMA lane maps (guide cdna_hip_programming.md §3):  A: lane l holds A[row=l&15][k=l>>4];
//    B: lane l holds B[k=l>>4][col=l&15];  D: reg r of lane l is D[row=(l>>4)+4r][col=l&15].
//  * blockIdx -> tile map is XCD aware: the 8 XCDs (private L2 each) get contiguous chunks of the
//    logical tile order, m-tiles fastest, so the m-tiles that share one streamed B panel (the (+/-) pair-packed
//    W_vvvv operands of the ladder: 3.2 GB each) run on one XCD at the same time and the panel is fetched from HBM once.
//  * the large tiles run the MODE 1 main loop (explicit ds_read_b64 fragment reads one k-step ahead, LDS stores spread behind
//    the MFMA rows, barrier before the last k-step; see the comment above it); DESIGN.md section 4 has the measurements.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "dev_ops.h"
#include "hip_common.h"
#include "grouped_launch.h"

namespace qemb {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));      // a pair read from global memory at 8-byte alignment (rows of an odd leading dimension)

// Test / tuning hooks (qemb_set_gemm_config, qemb_set_gemm_splitk).  Per calling HOST THREAD: a thread drives one execution context
// (stream), so a setter cannot change the tile choice of GEMMs dispatched -- or being captured into a hipGraph -- by another thread.
static thread_local int t_gemm_force_cfg = -1;
static thread_local int t_gemm_splitk_enabled = 1;
static thread_local int t_gemm_peers = 1;
static std::atomic<long long> g_gemm_flops{0};
int dev_gemm_flop_count(double* flops, int reset) { if (flops) *flops = (double)g_gemm_flops.load(); if (reset) g_gemm_flops = 0; return QEMB_OK; }
void dev_gemm_set_peers(int n) { t_gemm_peers = n > 1 ? n : 1; }
int dev_gemm_peers() { return t_gemm_peers; }
void dev_gemm_set_force_cfg(int cfg) { t_gemm_force_cfg = cfg; }
void dev_gemm_set_auto_splitk(int enabled) { t_gemm_splitk_enabled = enabled; }

struct GemmKArgs {
  const double* A; const double* B; double* C;
  long long lda, ldb, ldc, strideA, strideB, strideC;
  int M, N, K;
  int tiles_m, tiles_n;
  int ksplit, kchunk;      // split-K: blockIdx.y = batch * ksplit + slice; slice s covers k in [s*kchunk, (s+1)*kchunk)
  double alpha, beta;
  long long* cyc2;         // TAG == 2: prologue stamps
  long long* cyc;          // debugging (QEMB_GEMM_TRACE): per-workgroup shader-clock ticks, or nullptr
  int sb_m, sb_n;          // > 0: the tiles of an XCD's chunk are walked in super-blocks of sb_m x sb_n tiles (launch_cfg); 0: m-tiles fastest
  int a_slab; long long a_slab_skip;      // slab-aware rows of a !A_KC operand (GemmDesc::a_slab): row m is a_slab_skip * (m / a_slab) elements further on; 0: plain
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective XCD-contiguous remap (guide §5 "XCD swizzle must be bijective")
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

template <int BMN, int BK, bool KCONTIG>
struct LdsImage {
  static constexpr int PADK = 2;
  static constexpr int PADMN = (16 - (BMN % 32) + 32) % 32;
  static constexpr int LDK = BK + PADK;      // row stride (doubles) of the K-contiguous image
  static constexpr int LDMN = BMN + PADMN;   // row stride (doubles) of the M/N-contiguous image
  static constexpr int SIZE = KCONTIG ? BMN * LDK : BK * LDMN;
  __host__ __device__ static constexpr int off(int mn, int k) {
    return KCONTIG ? mn * LDK + k : k * LDMN + mn;
  }
};

// Stage one operand tile global -> registers (zero filled outside the matrix).
template <int BMN, int BK, bool KCONTIG, int VEC, int T, int NCH>
__device__ __forceinline__ void stage_load(double (&reg)[NCH][VEC], const double* __restrict__ P,
                                           long long ld, int mn0, int k0, int MN, int K, int tid, int slab = 0, long long slab_skip = 0) {
  constexpr int CPR = (KCONTIG ? BK : BMN) / VEC;  // chunks per contiguous row of the tile
  constexpr int TOTAL = BMN * BK / VEC;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int chunk = tid + c * T;
    if ((TOTAL % T != 0) && chunk >= TOTAL) { reg[c][0] = 0.0; reg[c][VEC - 1] = 0.0; continue; }
    const int r = chunk / CPR;             // K-contig: tile row (m/n) ; else: tile k
    const int cc = (chunk % CPR) * VEC;    // offset along the contiguous dim
    const int mn = KCONTIG ? r : cc;
    const int k = KCONTIG ? cc : r;
    const int gmn = mn0 + mn, gk = k0 + k;
    long long g = KCONTIG ? (long long)gmn * ld + gk : (long long)gk * ld + gmn;
    if (!KCONTIG && slab > 0) g += (long long)(gmn / slab) * slab_skip;
    if constexpr (VEC == 2) {
      // M/N-contiguous operands: the dispatch guarantees even extents / ld / alignment, so a chunk is entirely in or out.  K-contiguous operands may have an odd
      // K (and an odd ld: pairs at 8-byte alignment): the last pair of a row is then its last element alone
      const bool ok = (gmn < MN) && (gk < K);
      d2 v = {0.0, 0.0};
      if (KCONTIG) {
        if (ok && gk + 1 < K) { const d2u u = *reinterpret_cast<const d2u*>(P + g); v[0] = u[0]; v[1] = u[1]; }
        else if (ok) v[0] = P[g];
      } else if (ok) v = *reinterpret_cast<const d2*>(P + g);
      reg[c][0] = v[0];
      reg[c][VEC - 1] = v[1];
    } else {
      const bool ok = (gmn < MN) && (gk < K);
      reg[c][0] = ok ? P[g] : 0.0;
    }
  }
}

// Interior tiles: per-thread chunk pointers are set up once (stage_ptrs) and advanced by one k-tile per call (stage_load_fast) -- two
// integer adds per chunk in the main loop instead of the index arithmetic, bounds tests and branches of stage_load.  Rows beyond the
// matrix are clamped to its last row (their products land in rows / columns of the tile that are never stored); only a k-tail needs
// zeros, so the LAST tile of a slice still goes through stage_load.
template <int BMN, int BK, bool KCONTIG, int VEC, int T, int NCH>
__device__ __forceinline__ void stage_ptrs(const double* (&ptr)[NCH], const double* __restrict__ P, long long ld, int mn0, int k0, int MN, int tid, int slab = 0,
                                           long long slab_skip = 0) {
  constexpr int CPR = (KCONTIG ? BK : BMN) / VEC;
  constexpr int TOTAL = BMN * BK / VEC;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int chunk = tid + c * T;
    if ((TOTAL % T != 0) && chunk >= TOTAL) { ptr[c] = P; continue; }
    const int r = chunk / CPR, cc = (chunk % CPR) * VEC;
    const int mn = KCONTIG ? r : cc, k = KCONTIG ? cc : r;
    int gmn = mn0 + mn;
    const int last = KCONTIG ? MN - 1 : MN - VEC;    // last addressable row / (VEC == 2, mn contiguous: MN and gmn are even) last chunk start
    gmn = gmn < last ? gmn : (last > 0 ? last : 0);
    ptr[c] = KCONTIG ? P + (long long)gmn * ld + (k0 + k) : P + (long long)(k0 + k) * ld + gmn;
    if (!KCONTIG && slab > 0) ptr[c] += (long long)(gmn / slab) * slab_skip;
  }
}
template <int BMN, int BK, bool KCONTIG, int VEC, int T, int NCH>
__device__ __forceinline__ void stage_load_fast(double (&reg)[NCH][VEC], const double* (&ptr)[NCH], long long step, int tid) {
  // (threads whose chunk index runs past a tile that does not divide over the workgroup load from the operand's base instead -- a valid
  //  address; stage_store never writes that register.  No divergence here: a branch would make the compiler wait for the loads in flight.)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if constexpr (VEC == 2) {
      if constexpr (KCONTIG) { const d2u v = *reinterpret_cast<const d2u*>(ptr[c]); reg[c][0] = v[0]; reg[c][VEC - 1] = v[1]; }
      else { const d2 v = *reinterpret_cast<const d2*>(ptr[c]); reg[c][0] = v[0]; reg[c][VEC - 1] = v[1]; }
    } else {
      reg[c][0] = *ptr[c];
    }
    ptr[c] += ((BMN * BK / VEC) % T != 0 && (tid + c * T) >= BMN * BK / VEC) ? 0 : step;   // a select, not a branch
  }
}

template <int BMN, int BK, bool KCONTIG, int VEC, int T, int NCH>
__device__ __forceinline__ void stage_store(const double (&reg)[NCH][VEC], double* __restrict__ S,
                                            int tid) {
  using Img = LdsImage<BMN, BK, KCONTIG>;
  constexpr int CPR = (KCONTIG ? BK : BMN) / VEC;
  constexpr int TOTAL = BMN * BK / VEC;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int chunk = tid + c * T;
    if ((TOTAL % T != 0) && chunk >= TOTAL) continue;
    const int r = chunk / CPR;
    const int cc = (chunk % CPR) * VEC;
    const int mn = KCONTIG ? r : cc;
    const int k = KCONTIG ? cc : r;
    double* dst = S + Img::off(mn, k);
    if constexpr (VEC == 2) {
      d2 v = {reg[c][0], reg[c][VEC - 1]};
      *reinterpret_cast<d2*>(dst) = v;
    } else {
      dst[0] = reg[c][0];
    }
  }
}


// ---- MODE 1 main loop: explicit LDS fragment reads --------------------------------------------------------------------------------
// The compiler fuses neighbouring ds_read_b64 of the classic loop into ds_read2_b64: half the LDS rate (4 x 16-lane groups, banks mod
// 32) and, on the [row][BK+2] image, 2-way conflicts -- 16 LDS cycles for two fragment reads instead of 4 -- and it issues all reads
// of a k-step in one burst after the previous step's last MFMA, so both waves of a SIMD (in lockstep after the per-tile barrier) leave
// the matrix pipe idle while eight waves queue on the LDS.  Here every fragment is read by an explicit ds_read_b64 (never fused) and
// the reads run ONE k-step ahead: the B fragments are double buffered, each A fragment register is refilled right after the last MFMA
// that consumed it, and the per-tile barrier sits before the LAST k-step of the tile, whose fragments are by then in registers -- so
// the first reads of the next tile (issued during that last k-step) find their LDS buffer published and never wait.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ unsigned lds_byte_address(const double* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const double*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read_f64(double& dst, unsigned base) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b64 immediate offset");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(OFF));
}
// wait until at most N LDS operations of this wave are outstanding; x is "modified" so that its consumers are ordered behind the wait
template <int N>
__device__ __forceinline__ void lds_wait(double& x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
}

// one chunk (index C of NCH) of stage_store: lets the MODE 1 loop spread the LDS stores of the next tile between its MFMA rows
template <int BMN, int BK, bool KCONTIG, int VEC, int T, int NCH, int C>
__device__ __forceinline__ void stage_store_chunk(const double (&reg)[NCH][VEC], double* __restrict__ S, int tid) {
  using Img = LdsImage<BMN, BK, KCONTIG>;
  constexpr int CPR = (KCONTIG ? BK : BMN) / VEC;
  constexpr int TOTAL = BMN * BK / VEC;
  const int chunk = tid + C * T;
  if ((TOTAL % T != 0) && chunk >= TOTAL) return;
  const int r = chunk / CPR;
  const int cc = (chunk % CPR) * VEC;
  const int mn = KCONTIG ? r : cc;
  const int k = KCONTIG ? cc : r;
  double* dst = S + Img::off(mn, k);
  if constexpr (VEC == 2) {
    d2 v = {reg[C][0], reg[C][VEC - 1]};
    *reinterpret_cast<d2*>(dst) = v;
  } else {
    dst[0] = reg[C][0];
  }
}

// the value held by the neighbouring lane (lane ^ 1): two DPP moves (quad_perm [1,0,3,2]) on the halves of the double
__device__ __forceinline__ double lane_swap_neighbour(double x) {
  const unsigned long long u = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0xB1, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0xB1, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// TAG does not change the code: it gives a call site its own kernel symbol so that profiles (rocprofv3 --stats, PMC) of the
// pp-ladder are not mixed with other users of the same tile shape.
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool A_KC, bool B_KC, int VEC, int TAG = 0, int MODE = 0>
__device__ __forceinline__ void dgemm_mfma_body(const uint3 BID, const uint3 GDIM, GemmKArgs g) {
  const long long t_start = g.cyc ? (long long)__builtin_amdgcn_s_memtime() : 0;
  constexpr int BM = WM * 16 * WAVES_M;
  constexpr int BN = WN * 16 * WAVES_N;
  constexpr int T = WAVES_M * WAVES_N * 64;
  using ImgA = LdsImage<BM, BK, A_KC>;
  using ImgB = LdsImage<BN, BK, B_KC>;
  constexpr int NCH_A = (BM * BK / VEC + T - 1) / T;
  constexpr int NCH_B = (BN * BK / VEC + T - 1) / T;
  static_assert(BK % 4 == 0, "BK must be a multiple of the MFMA K (4)");

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sA0 = smem;
  double* sB0 = sA0 + ImgA::SIZE;
  double* sA1 = sB0 + ImgB::SIZE;
  double* sB1 = sA1 + ImgA::SIZE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int ntiles = g.tiles_m * g.tiles_n;
  const int L = xcd_remap(BID.x, ntiles);
  int tm = L % g.tiles_m, tn = L / g.tiles_m;
  if (g.sb_m > 0) {
    // 2-D walk inside an XCD's chunk (round 4): the 32 workgroups an XCD runs at a time form an sb_m x sb_n block of output tiles, so that
    // they share sb_m A panels and sb_n B panels through the XCD's L2 instead of 32 A panels and one B panel -- the (ov)^3 ring products
    // (32 x 16 tiles of 128 x 256) drew 2.4 GB per call through the fabric for 0.38 GB of operands (profiles/r04_hbm_pmc*.json, all_kernels)
    const int per = g.sb_m * g.sb_n, sb = L / per, w = L - sb * per, nsbm = g.tiles_m / g.sb_m;
    tm = (sb % nsbm) * g.sb_m + w % g.sb_m;
    tn = (sb / nsbm) * g.sb_n + w / g.sb_m;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const long long bz = BID.y / g.ksplit;
  const int kslice = BID.y % g.ksplit;
  const int kbeg = kslice * g.kchunk;
  const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
  const double* __restrict__ A = g.A + bz * g.strideA;
  const double* __restrict__ B = g.B + bz * g.strideB;
  // with split-K every slice writes its own partial (C = workspace [batch][slice][M][N]); combined by splitk_reduce
  double* __restrict__ C = g.C + (long long)BID.y * g.strideC;

  d4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // Operand staging is one tile deep for A and TWO tiles deep for B by default (DB): the B tile of step kt + 2 is requested before the
  // MFMAs of step kt and only stored to LDS at the end of step kt + 1 -- a full tile more for the HBM latency; A (re-read by every
  // column tile, L2 resident) stays one deep where the accumulators leave no registers for a second slot (the 7 x 2 and 4 x 4 wave tiles).
  // (MODE 1 refills the one register set right after its LDS stores, a full tile before the next ones: no second set needed)
  constexpr int DA = (MODE == 0 && VEC == 2 && (WM * WN <= 8 || (WM == 6 && WN == 2))) ? 2 : 1;   // small wave tiles and the 6 x 2 ladder tile have the registers
  constexpr int DB = (MODE == 0 && VEC == 2 && WM < 14) ? 2 : 1;    // (the scalar-load variants and the 14 x 1 wave tile would spill)
  double ra[DA][NCH_A][VEC], rb[DB][NCH_B][VEC];
  const int nk = (kend - kbeg + BK - 1) / BK;

  const double* pa[NCH_A];
  const double* pb[NCH_B];
  stage_ptrs<BM, BK, A_KC, VEC, T, NCH_A>(pa, A, g.lda, m0, kbeg + BK, g.M, tid, g.a_slab, g.a_slab_skip);   // tile 1 is the first one the pointer-bump loader fetches
  stage_ptrs<BN, BK, B_KC, VEC, T, NCH_B>(pb, B, g.ldb, n0, kbeg + BK, g.N, tid);
  const long long step_a = A_KC ? (long long)BK : (long long)BK * g.lda;
  const long long step_b = B_KC ? (long long)BK : (long long)BK * g.ldb;
  // tile tt of the slice into a register slot: interior tiles through the pointer-bump loader, the last one (k-tail -> zeros) guarded
  // (the scalar-load variants, VEC == 1, keep the guarded loader throughout: twice the chunks, so the pointer arrays would spill)
  constexpr bool FAST = (VEC == 2);
  auto fetch_a = [&](double (&r)[NCH_A][VEC], int tt) {
    if (FAST && tt + 1 < nk) stage_load_fast<BM, BK, A_KC, VEC, T, NCH_A>(r, pa, step_a, tid);
    else if (tt < nk) stage_load<BM, BK, A_KC, VEC, T, NCH_A>(r, A, g.lda, m0, kbeg + tt * BK, g.M, kend, tid, g.a_slab, g.a_slab_skip);
  };
  auto fetch_b = [&](double (&r)[NCH_B][VEC], int tt) {
    if (FAST && tt + 1 < nk) stage_load_fast<BN, BK, B_KC, VEC, T, NCH_B>(r, pb, step_b, tid);
    else if (tt < nk) stage_load<BN, BK, B_KC, VEC, T, NCH_B>(r, B, g.ldb, n0, kbeg + tt * BK, g.N, kend, tid);
  };

  // prologue: tile 0 -> LDS buffer 0; with a two-deep operand, tile 1 is already requested
  long long pst[4] = {0, 0, 0, 0};     // TAG == 2: prologue stamps
  if constexpr (TAG == 2) pst[0] = __builtin_amdgcn_s_memtime();
  stage_load<BM, BK, A_KC, VEC, T, NCH_A>(ra[0], A, g.lda, m0, kbeg, g.M, kend, tid, g.a_slab, g.a_slab_skip);
  stage_load<BN, BK, B_KC, VEC, T, NCH_B>(rb[0], B, g.ldb, n0, kbeg, g.N, kend, tid);
  stage_store<BM, BK, A_KC, VEC, T, NCH_A>(ra[0], sA0, tid);
  stage_store<BN, BK, B_KC, VEC, T, NCH_B>(rb[0], sB0, tid);
  if constexpr (TAG == 2) pst[1] = __builtin_amdgcn_s_memtime();
  if (DA == 2 || MODE == 1) fetch_a(ra[DA - 1], 1);
  if (DB == 2 || MODE == 1) fetch_b(rb[DB - 1], 1);
  if constexpr (TAG == 2) pst[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if constexpr (TAG == 2) pst[3] = __builtin_amdgcn_s_memtime();

  const int fr = lane & 15, fk = lane >> 4;
  if constexpr (MODE == 1) {
    static_assert(VEC == 2, "MODE 1 uses the vector staging path");
    constexpr int NKS = BK / 4;
    static_assert(NKS >= 2 && NKS % 2 == 0, "the B fragment sets alternate per k-step");
    // LDS byte addresses of this lane's fragment origin in the two buffers; fragment (row block i, k-step ks) is a constant offset away
    const unsigned ldsA[2] = {lds_byte_address(sA0 + ImgA::off(wm * WM * 16 + fr, fk)), lds_byte_address(sA1 + ImgA::off(wm * WM * 16 + fr, fk))};
    const unsigned ldsB[2] = {lds_byte_address(sB0 + ImgB::off(wn * WN * 16 + fr, fk)), lds_byte_address(sB1 + ImgB::off(wn * WN * 16 + fr, fk))};
    double a[WM], b[2][WN];
    // fragments of (tile 0, k-step 0): B first, then A row by row (LDS returns in order)
    static_for<0, WN>([&](auto j) { lds_read_f64<ImgB::off(decltype(j)::value * 16, 0) * 8>(b[0][decltype(j)::value], ldsB[0]); });
    static_for<0, WM>([&](auto i) { lds_read_f64<ImgA::off(decltype(i)::value * 16, 0) * 8>(a[decltype(i)::value], ldsA[0]); });
    // one k-step of tile parity P: MFMAs of k-step KS on the fragments in registers, reads for the following k-step (same tile, or k-step
    // 0 of the next tile from the other buffer) issued underneath.  The very last k-step of the slice issues them too -- they read
    // whatever the other buffer holds and are never used -- so that the loop body is one straight instruction stream (a second copy of
    // the k-step without the reads makes the register allocator shuffle the accumulators between the two copies and spill).
    long long st_work = 0, st_bar = 0, st_last = 0, st_t0 = 0;   // TAG == 2 (diagnostic instantiation): per-wave s_memtime stamps
    auto kstep = [&](auto ks_c, auto p_c, auto&& after_row) {
      constexpr int KS = decltype(ks_c)::value, P = decltype(p_c)::value;
      constexpr int S = KS & 1;                                   // B fragment set in use
      constexpr int KSn = (KS + 1) % NKS, Pn = (KS + 1 < NKS) ? P : 1 - P;
      static_for<0, WN>([&](auto j) { lds_read_f64<ImgB::off(decltype(j)::value * 16, KSn * 4) * 8>(b[S ^ 1][decltype(j)::value], ldsB[Pn]); });
      static_for<0, WM>([&](auto i) {
        constexpr int I = decltype(i)::value;
        // LDS operations issued after the read of a[I]: the later rows of its own k-step, the next B set and the refills of the rows
        // already consumed in this k-step -- WM - 1 + WN in all, whatever I is (LDS stores slipped in between only make the wait stricter)
        constexpr int NW = WM + WN - 1;
        if constexpr (I == 0) static_for<0, WN>([&](auto j) { lds_wait<NW>(b[S][decltype(j)::value]); });
        lds_wait<NW>(a[I]);
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[I][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], b[S][j], acc[I][j], 0, 0, 0);
        if constexpr (!((TAG == 4 || TAG == 5) && (I & 1)))     // (ablation instantiations 4 / 5: every other A fragment is not re-read)
          lds_read_f64<ImgA::off(I * 16, KSn * 4) * 8>(a[I], ldsA[Pn]);
        after_row(ks_c, i);
      });
    };
    // The LDS stores of tile kt + 1 (other buffer: its last readers passed the previous barrier) are spread one chunk at a time behind
    // the MFMA rows of k-step NKS - 2 instead of one burst in front of the barrier, where eight waves queue on the LDS store path with
    // the matrix pipe idle.  Past the end of the slice they store stale registers into a buffer nobody reads.
    constexpr int NCH = NCH_A + NCH_B, NSLOT = WM;
    auto tile = [&](auto parity, int kt) {
      constexpr int P = decltype(parity)::value;
      double* nA = P ? sA0 : sA1;
      double* nB = P ? sB0 : sB1;
      auto stores = [&](auto ks_c, auto i_c) {
        constexpr int KS = decltype(ks_c)::value, I = decltype(i_c)::value;
        if constexpr (KS == NKS - 2) {
          constexpr int slot = I;
          // chunks c with c * NSLOT / NCH == slot (several per slot when there are more chunks than rows)
          static_for<0, NCH>([&](auto c_c) {
            constexpr int C = decltype(c_c)::value;
            if constexpr ((C * NSLOT) / NCH == slot) {
              if constexpr (C < NCH_A) stage_store_chunk<BM, BK, A_KC, VEC, T, NCH_A, C>(ra[0], nA, tid);
              else stage_store_chunk<BN, BK, B_KC, VEC, T, NCH_B, C - NCH_A>(rb[0], nB, tid);
            }
          });
        }
      };
      long long ts0 = 0;
      if constexpr (TAG == 2) ts0 = __builtin_amdgcn_s_memtime();
      static_for<0, NKS - 1>([&](auto ks) { kstep(ks, parity, stores); });
      // the staging registers are free again: request tile kt + 2 now -- it is stored during k-step NKS - 2 of the NEXT tile, NKS - 1
      // k-steps of MFMAs away (the loads behind the vmcnt wait of this tile's stores were issued that long ago)
      if constexpr (TAG != 3 && TAG != 5) {     // (ablation instantiations 3 / 5: no global loads in the main loop)
        fetch_a(ra[0], kt + 2);
        fetch_b(rb[0], kt + 2);
      }
      long long ts1 = 0;
      if constexpr (TAG == 2) ts1 = __builtin_amdgcn_s_memtime();
      __syncthreads();   // (also drains this wave's fragment reads of k-step NKS - 1: nobody reads this tile's buffer after the barrier)
      if constexpr (TAG == 2) { const long long ts2 = __builtin_amdgcn_s_memtime(); st_work += ts1 - ts0; st_bar += ts2 - ts1; st_t0 = ts2; }
      kstep(std::integral_constant<int, NKS - 1>{}, parity, stores);
      if constexpr (TAG == 2) st_last += __builtin_amdgcn_s_memtime() - st_t0;
    };
    for (int kt = 0; kt < nk; kt += 2) {
      tile(std::integral_constant<int, 0>{}, kt);
      if (kt + 1 < nk) tile(std::integral_constant<int, 1>{}, kt + 1);
    }
    if constexpr (TAG == 2) {
      if (g.cyc && lane == 0) {     // [workgroup][wave][3]: cycles in k-steps 0..NKS-2 (+ LDS stores), at the barrier (+ fetch issue), in the last k-step
        long long* o = g.cyc + (((long long)BID.y * GDIM.x + BID.x) * (WAVES_M * WAVES_N) + wave) * 4;
        o[0] = st_work; o[1] = st_bar; o[2] = st_last; o[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;   // o[3]: kernel entry -> end of the main loop
        if (g.cyc2) { long long* q = g.cyc2 + (((long long)BID.y * GDIM.x + BID.x) * (WAVES_M * WAVES_N) + wave) * 4; q[0] = pst[0] - t_start; q[1] = pst[1] - pst[0]; q[2] = pst[2] - pst[1]; q[3] = pst[3] - pst[2]; }
      }
    }
    // the run-ahead reads of the final k-step are still in flight: let them land before the fragment registers are reused
    static_for<0, WN>([&](auto j) { lds_wait<0>(b[0][decltype(j)::value]); lds_wait<0>(b[1][decltype(j)::value]); });
    static_for<0, WM>([&](auto i) { lds_wait<0>(a[decltype(i)::value]); });
  } else {
  // one k-tile; P = kt & 1 is a compile-time parity so that every register slot index is static
    auto step = [&](auto parity, int kt) {
      constexpr int P = decltype(parity)::value;
      const double* sA = P ? sA1 : sA0;
      const double* sB = P ? sB1 : sB0;
      double* nA = P ? sA0 : sA1;
      double* nB = P ? sB0 : sB1;
      // requests for the tile DA / DB steps ahead (tile kt + 2 lands in slot P, tile kt + 1 of a one-deep operand in slot 0)
      fetch_a(ra[DA == 2 ? P : 0], kt + DA);
      fetch_b(rb[DB == 2 ? P : 0], kt + DB);
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        const int kk = ks * 4 + fk;
        double a[WM], b[WN];
        // B fragments first, then A row by row: LDS returns in order, so the MFMAs of row i only wait for a[0..i] (partial lgkmcnt)
        // and the later fragment reads land under the earlier rows' MFMAs
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = sB[ImgB::off((wn * WN + j) * 16 + fr, kk)];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = sA[ImgA::off((wm * WM + i) * 16 + fr, kk)];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nk) {      // tile kt + 1 (requested one or two steps ago) -> the other LDS buffer
        stage_store<BM, BK, A_KC, VEC, T, NCH_A>(ra[DA == 2 ? 1 - P : 0], nA, tid);
        stage_store<BN, BK, B_KC, VEC, T, NCH_B>(rb[DB == 2 ? 1 - P : 0], nB, tid);
      }
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      step(std::integral_constant<int, 0>{}, kt);
      if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
    }

  }

  // epilogue: D reg r of lane l -> row (l>>4)+4r, col l&15 of the 16x16 tile
  const double alpha = g.alpha, beta = g.beta;
  // A lane owns four ROWS of one column: the natural store is four 8-byte stores per 16 x 16 tile, and short-K products (the quarter
  // transforms, K ~ 220: 14 k-tiles per 224 x 128 tile) are then bound by the ISSUE of their 56 store instructions per lane, not by
  // bytes.  Neighbouring lanes (columns 2c, 2c+1) trade two values each -- the even lane keeps rows +0/+4, the odd lane rows +8/+12 --
  // so that every lane stores two 16-byte pairs: half the store instructions, 1 KB instead of 512 B per wave instruction.
  const bool wide = ((g.ldc & 1) == 0) && ((g.N & 1) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  if (wide) {
    const bool odd = (fr & 1) != 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const d4 v = acc[i][j];
        const double r0 = lane_swap_neighbour(odd ? v[0] : v[2]), r1 = lane_swap_neighbour(odd ? v[1] : v[3]);
        const d2 pa = odd ? d2{r0, v[2]} : d2{v[0], r0};
        const d2 pb = odd ? d2{r1, v[3]} : d2{v[1], r1};
        const int col = n0 + (wn * WN + j) * 16 + (fr & ~1);
        const int row = m0 + (wm * WM + i) * 16 + fk + (odd ? 8 : 0);
        if (col < g.N) {
          if (row < g.M) {
            d2* p = reinterpret_cast<d2*>(C + (long long)row * g.ldc + col);
            d2 o = {alpha * pa[0], alpha * pa[1]};
            if (beta != 0.0) { const d2 c = *p; o[0] += beta * c[0]; o[1] += beta * c[1]; }
            *p = o;
          }
          if (row + 4 < g.M) {
            d2* p = reinterpret_cast<d2*>(C + (long long)(row + 4) * g.ldc + col);
            d2 o = {alpha * pb[0], alpha * pb[1]};
            if (beta != 0.0) { const d2 c = *p; o[0] += beta * c[0]; o[1] += beta * c[1]; }
            *p = o;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n0 + (wn * WN + j) * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * WM + i) * 16 + fk + 4 * r;
          if (row < g.M && col < g.N) {
            double* p = C + (long long)row * g.ldc + col;
            double v = alpha * acc[i][j][r];
            if (beta != 0.0) v += beta * (*p);
            *p = v;
          }
        }
      }
    }
  }
  if (TAG != 2 && g.cyc && threadIdx.x == 0) g.cyc[(long long)BID.y * GDIM.x + BID.x] = (long long)__builtin_amdgcn_s_memtime() - t_start;
  if (TAG == 2 && g.cyc && lane == 0) g.cyc[(long long)GDIM.x * GDIM.y * (WAVES_M * WAVES_N) * 4 + ((long long)BID.y * GDIM.x + BID.x) * (WAVES_M * WAVES_N) + wave] = (long long)__builtin_amdgcn_s_memtime() - t_start;   // whole wave lifetime
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool A_KC, bool B_KC, int VEC, int TAG = 0, int MODE = 0>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64, (WM * WN <= 16) ? 2 : 1)
    dgemm_mfma_kernel(GemmKArgs g) {
  dgemm_mfma_body<WM, WN, WAVES_M, WAVES_N, BK, A_KC, B_KC, VEC, TAG, MODE>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), g);
}

// C[b][m][n] = alpha * sum_s ws[b][s][m][n] + beta * C   (fixed summation order: deterministic)
__device__ __forceinline__ void splitk_reduce_kernel_body(const uint3 BID, const uint3 GDIM, const double* __restrict__ ws, int S, long long M, long long N,
                                                            double* __restrict__ C, long long ldc, long long strideC,
                                                            double alpha, double beta) {
  const long long mn = M * N;
  const long long b = BID.y;
  for (long long t = (long long)BID.x * blockDim.x + threadIdx.x; t < mn; t += (long long)GDIM.x * blockDim.x) {
    double acc = 0.0;
    const double* p = ws + b * S * mn + t;
    int s = 0;
    for (; s + 8 <= S; s += 8) {      // eight slabs' loads in flight, added in slab order (a counted loop waits for every load in turn)
      double x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = p[(long long)(s + q) * mn];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += x[q];
    }
    for (; s < S; ++s) acc += p[(long long)s * mn];
    const long long m = t / N, n = t - m * N;
    double* c = C + b * strideC + m * ldc + n;
    *c = (beta != 0.0) ? alpha * acc + beta * (*c) : alpha * acc;
  }
}
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const double* __restrict__ ws, int S, long long M, long long N,
                                                            double* __restrict__ C, long long ldc, long long strideC,
                                                            double alpha, double beta) { splitk_reduce_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), ws, S, M, N, C, ldc, strideC, alpha, beta); }
// Few outputs, many slabs (the o x o / v x v shaped intermediates with K = o v^2 split several hundred ways): one WAVE per
// output element, lane l sums slabs l, l+64, ... and the 64 partial sums are combined in a fixed butterfly order.
__device__ __forceinline__ void splitk_reduce_wave_kernel_body(const uint3 BID, const uint3 GDIM, const double* __restrict__ ws, int S, long long M, long long N,
                                                                 double* __restrict__ C, long long ldc, long long strideC,
                                                                 double alpha, double beta) {
  const long long mn = M * N;
  const long long b = BID.y;
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)BID.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)GDIM.x * blockDim.x) >> 6;
  for (long long t = wave; t < mn; t += nwaves) {
    double acc = 0.0;
    const double* p = ws + b * S * mn + t;
    for (int s = lane; s < S; s += 64) acc += p[(long long)s * mn];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
      const long long m = t / N, n = t - m * N;
      double* c = C + b * strideC + m * ldc + n;
      *c = (beta != 0.0) ? alpha * acc + beta * (*c) : alpha * acc;
    }
  }
}
__global__ void __launch_bounds__(256) splitk_reduce_wave_kernel(const double* __restrict__ ws, int S, long long M, long long N,
                                                                 double* __restrict__ C, long long ldc, long long strideC,
                                                                 double alpha, double beta) { splitk_reduce_wave_kernel_body(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), ws, S, M, N, C, ldc, strideC, alpha, beta); }

double* gemm_workspace(size_t bytes);   // dev_ops_hip.hip
static long long* g_gemm_cyc = nullptr;          // QEMB_GEMM_TRACE: per-workgroup tick buffer of the traced launch
static long long g_gemm_cyc_cap = 0, g_gemm_cyc_blocks = 0;
static bool g_gemm_cyc_on = false;

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool A_KC, bool B_KC, int VEC, int TAG = 0, int MODE = 0>
static int launch_cfg(const GemmDesc& d, hipStream_t s) {
  constexpr int BM = WM * 16 * WAVES_M, BN = WN * 16 * WAVES_N;
  using ImgA = LdsImage<BM, BK, A_KC>;
  using ImgB = LdsImage<BN, BK, B_KC>;
  GemmKArgs g{};      // (padding zeroed: recorded launches are compared byte by byte, dev_tape_equal)
  g.A = d.A; g.B = d.B; g.C = d.C;
  g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
  g.strideA = d.strideA; g.strideB = d.strideB; g.strideC = d.strideC;
  g.M = (int)d.M; g.N = (int)d.N; g.K = (int)d.K;
  g.tiles_m = (int)((d.M + BM - 1) / BM);
  g.tiles_n = (int)((d.N + BN - 1) / BN);
  g.alpha = d.alpha; g.beta = d.beta;
  g.cyc = nullptr; g.cyc2 = nullptr;
  g.sb_m = g.sb_n = 0;
  g.a_slab = 0; g.a_slab_skip = 0;
  if (d.a_slab > 0) {
    if (A_KC || d.batch != 1 || d.a_slab > 0x3fffffff || (VEC == 2 && (d.a_slab % 2 || d.a_slab_skip % 2))) { set_error("dev_gemm: a_slab needs a !a_kcontig A operand, batch = 1 and (16-byte loads) an even slab"); return QEMB_ERR_ARG; }
    g.a_slab = (int)d.a_slab; g.a_slab_skip = d.a_slab_skip;
  }
  {
    // super-blocks of 32 tiles (the workgroups one XCD runs at a time at one workgroup per CU): the shape that moves the fewest operand bytes
    // per k-step, sb_m x |A tile| + sb_n x |B tile|, among the shapes that divide the tile grid (QEMB_GEMM_SB=0: the m-fastest walk, for A/B runs)
    static const bool sb_on = !(std::getenv("QEMB_GEMM_SB") && std::atoi(std::getenv("QEMB_GEMM_SB")) == 0);
    if (sb_on && d.batch == 1 && d.ksplit <= 1 && g.tiles_m >= 8 && g.tiles_n >= 4 && ((long long)g.tiles_m * g.tiles_n) % 256 == 0) {
      long long best = -1;
      for (int sm = 1; sm <= 32; sm *= 2) {
        const int sn = 32 / sm;
        if (g.tiles_m % sm || g.tiles_n % sn) continue;
        const long long cost = (long long)sm * BM + (long long)sn * BN;
        if (best < 0 || cost < best) { best = cost; g.sb_m = sm; g.sb_n = sn; }
      }
      if (g.sb_m == 32) g.sb_m = g.sb_n = 0;      // that is the m-fastest walk already
    }
  }
  // split-K when the output has too few tiles to occupy 256 CUs but K is long (the o x v, o x o, v x v shaped
  // CCSD intermediates contract over o*v^2 ... v^2 indices)
  g.ksplit = 1; g.kchunk = g.K > 0 ? g.K : 1;
  const long long tiles = (long long)g.tiles_m * g.tiles_n * d.batch;
  if (d.keep_slabs && (d.ksplit <= 1 || d.alpha != 1.0 || d.beta != 0.0 || d.batch != 1)) { set_error("dev_gemm: keep_slabs needs ksplit > 1, alpha = 1, beta = 0, batch = 1"); return QEMB_ERR_ARG; }
  if (d.ksplit > 1) {
    long long chunk = (d.K + d.ksplit - 1) / d.ksplit;
    chunk = (chunk + 31) / 32 * 32;
    const long long S = (d.K + chunk - 1) / chunk;
    if (S > 1 && S * d.batch <= 65535) { g.ksplit = (int)S; g.kchunk = (int)chunk; }
  } else if (d.ksplit == 0 && t_gemm_splitk_enabled && tiles < 256 && d.K >= 1024) {      // (ksplit < 0: the caller wants NO split -- its product runs beside others that fill the chip)
    long long S = (768 + tiles - 1) / tiles;
    if (S > d.K / 256) S = d.K / 256;
    if (S > 1) {
      long long chunk = (d.K + S - 1) / S;
      chunk = (chunk + 31) / 32 * 32;
      S = (d.K + chunk - 1) / chunk;
      if (S > 1 && S * d.batch <= 65535) { g.ksplit = (int)S; g.kchunk = (int)chunk; }
    }
  }
  if (d.keep_slabs) {      // the slices' partial products go straight to the caller's slabs (gemm_slab_count(K, ksplit) of them, also when that is one)
    if (g.ksplit != gemm_slab_count(d.K, d.ksplit)) { set_error("dev_gemm: slab count mismatch"); return QEMB_ERR_ARG; }
    g.ldc = d.N; g.strideC = d.M * d.N;
  } else if (g.ksplit > 1) {
    double* ws = gemm_workspace(sizeof(double) * (size_t)d.batch * g.ksplit * d.M * d.N);
    if (!ws) return QEMB_ERR_ALLOC;
    g.C = ws; g.ldc = d.N; g.strideC = d.M * d.N; g.alpha = 1.0; g.beta = 0.0;
  }
  const size_t lds = 2 * (size_t)(ImgA::SIZE + ImgB::SIZE) * sizeof(double);
  auto kern = dgemm_mfma_kernel<WM, WN, WAVES_M, WAVES_N, BK, A_KC, B_KC, VEC, TAG, MODE>;
  static std::atomic<bool> attr_set{false};   // benign if two threads both set the attribute once
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)(d.batch * g.ksplit), 1);
  dim3 block(WAVES_M * WAVES_N * 64, 1, 1);
  if (g_gemm_cyc_on && g_gemm_cyc && (long long)grid.x * grid.y * (TAG == 2 ? 72 : 1) <= g_gemm_cyc_cap) { g.cyc = g_gemm_cyc; g_gemm_cyc_blocks = (long long)grid.x * grid.y; if (TAG == 2) g.cyc2 = g_gemm_cyc + (long long)grid.x * grid.y * (WAVES_M * WAVES_N) * 5; }
  hipLaunchKernelGGL(kern, grid, block, lds, s, g);
  if (g.ksplit > 1 && !d.keep_slabs) {
    const long long mn = d.M * d.N;
    if (g.ksplit >= 128 && mn <= 65536) {
      const unsigned gx = (unsigned)((mn + 3) / 4 < 4096 ? (mn + 3) / 4 : 4096);      // 4 waves per block, one wave per element
      hipLaunchKernelGGL(splitk_reduce_wave_kernel, dim3(gx, (unsigned)d.batch), dim3(256), 0, s, (const double*)g.C, g.ksplit,
                         (long long)d.M, (long long)d.N, d.C, (long long)d.ldc, (long long)d.strideC, d.alpha, d.beta);
    } else {
      const unsigned gx = (unsigned)((mn + 255) / 256 < 2048 ? (mn + 255) / 256 : 2048);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gx, (unsigned)d.batch), dim3(256), 0, s, (const double*)g.C, g.ksplit,
                         (long long)d.M, (long long)d.N, d.C, (long long)d.ldc, (long long)d.strideC, d.alpha, d.beta);
    }
  }
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int TAG = 0, int MODE = 0>
static int launch_layout(const GemmDesc& d, hipStream_t s, bool vec2) {
  const bool a = d.a_kcontig != 0, b = d.b_kcontig != 0;
  // (the scalar-load variants keep the classic main loop)
#define QEMB_GEMM_CASE(AK, BKC)                                                                \
  if (a == AK && b == BKC)                                                                      \
    return vec2 ? launch_cfg<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 2, TAG, MODE>(d, s)          \
                : launch_cfg<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 1, TAG, 0>(d, s);
  QEMB_GEMM_CASE(true, true)
  QEMB_GEMM_CASE(true, false)
  QEMB_GEMM_CASE(false, true)
  QEMB_GEMM_CASE(false, false)
#undef QEMB_GEMM_CASE
  return QEMB_ERR_ARG;
}

// 16-byte loads of an operand tile: an M/N-contiguous operand needs pairs that are entirely inside or outside the matrix and 16-byte aligned (even extent, ld,
// batch stride; aligned base); a K-contiguous one takes any K and ld -- its pairs are read at 8-byte alignment and the odd last element of a row alone (round 5:
// fragments with an odd n_occ n_virt ran the 8-byte kernels, and could not share a grouped launch with their even neighbours of a lock-step sweep)
static bool operand_vec2_ok(const double* p, int64_t ld, int64_t stride, int64_t contig_extent, bool kcontig) {
  static const bool relaxed = !(std::getenv("QEMB_GEMM_VEC2_ODD") && std::atoi(std::getenv("QEMB_GEMM_VEC2_ODD")) == 0);      // (0: the round-4 rule, for A/B runs)
  if (kcontig && relaxed) return (reinterpret_cast<uintptr_t>(p) & 7) == 0;
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 2 == 0) && (stride % 2 == 0) &&
         (contig_extent % 2 == 0);
}



// ---- calibration: v_mfma_f64_16x16x4_f64 issue rate of the whole chip, registers only (no LDS, no memory traffic) -----------
// An UPPER bound for the tiled GEMMs: 16 independent accumulator chains per wave (the 7 x 2 ladder tile has 14, the 4 x 4 tile 16),
// so no MFMA ever waits for the previous result of its own chain; operands differ per lane and per chain (not the constant operands
// a power-saving data path could exploit); 2 waves per SIMD by default (8-wave workgroups of the ladder tile: 2 per SIMD).
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f64_peak_kernel(double* out, int iters, double seed) {
  d4 acc[NACC];
  double a[NACC], b[4];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    a[i] = seed * (1.0 + 0.37 * i) + 1e-3 * (double)((threadIdx.x * 2654435761u + i * 40503u) & 1023u) - 0.5;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = seed * (0.5 - 0.21 * j) + 1e-3 * (double)((threadIdx.x * 40503u + j * 2654435761u) & 1023u) - 0.5;
  // (inline asm with the accumulator tied to itself: through the builtin the compiler copies all 128 accumulator registers
  //  between VGPRs and AGPRs around every trip of this loop -- 256 moves per 16 MFMAs)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i & 3]));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;   // keep the accumulators live
}
int dev_mfma_f64_peak(int iters, int blocks_per_cu, double* tflops) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  constexpr int NACC = 16;
  double* out = nullptr;
  HIP_TRY(hipMalloc((void**)&out, 64));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(mfma_f64_peak_kernel<NACC>, dim3(grid), dim3(256), 0, s, out, 64, 0.731);   // warm-up
  HIP_TRY(hipEventRecord(e0, s));
  hipLaunchKernelGGL(mfma_f64_peak_kernel<NACC>, dim3(grid), dim3(256), 0, s, out, iters, 0.731);
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * grid;   // NACC MFMAs x 4 waves per block
  *tflops = flop / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  return QEMB_OK;
}

static int dev_gemm_dispatch(const GemmDesc& d);

// QEMB_GEMM_TRACE=1: every product is timed on its own (events + a stream sync) and logged -- a debugging aid, not a mode to run in
// One launch timed on its own, with the shader clock read back: every workgroup records its s_memtime ticks; the sum of the ticks over
// (256 CUs x kernel time) is the sustained clock when exactly one workgroup is resident per CU (the 8-wave ladder tiles) and a multiple
// of it when several are.  Syncs the stream -- a measuring aid, never on a timed path.
int dev_gemm_probe(const GemmDesc& d, double* ms_out, double* ghz_out, long long* workgroups) {
  hipEvent_t t0, t1;
  if (dev_capturing() || hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) { set_error("dev_gemm_probe: cannot time this launch"); return QEMB_ERR_DEVICE; }
  hipStream_t s = hip_stream();
  float ms = 0.f;
  if (!g_gemm_cyc) { g_gemm_cyc_cap = 1 << 20; if (hipMalloc((void**)&g_gemm_cyc, sizeof(long long) * g_gemm_cyc_cap) != hipSuccess) g_gemm_cyc = nullptr; }
  g_gemm_cyc_blocks = 0;
  g_gemm_cyc_on = true;
  (void)hipEventRecord(t0, s);
  const int rc = dev_gemm_dispatch(d);
  (void)hipEventRecord(t1, s);
  (void)hipEventSynchronize(t1);
  g_gemm_cyc_on = false;
  (void)hipEventElapsedTime(&ms, t0, t1);
  (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  double ghz = 0.0;
  if (g_gemm_cyc && g_gemm_cyc_blocks > 0 && ms > 0) {
    std::vector<long long> h((size_t)g_gemm_cyc_blocks);
    if (hipMemcpy(h.data(), g_gemm_cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      double sum = 0; for (long long c : h) sum += (double)c;
      ghz = sum / 256.0 / (ms * 1e6);
    }
  }
  if (ms_out) *ms_out = ms;
  if (ghz_out) *ghz_out = ghz;
  if (workgroups) *workgroups = g_gemm_cyc_blocks;
  return rc;
}

// One launch of a diagnostic (TAG 2) tile configuration; returns the per-wave stamp sums averaged over all waves:
// out[0] = cycles in k-steps 0..NKS-2, out[1] = cycles from there to past the barrier, out[2] = cycles in the last k-step, out[3] = kernel ms
int dev_gemm_stamps(const GemmDesc& d, int waves_per_wg, double* out) {
  hipStream_t s = hip_stream();
  if (!g_gemm_cyc) { g_gemm_cyc_cap = 1 << 20; if (hipMalloc((void**)&g_gemm_cyc, sizeof(long long) * g_gemm_cyc_cap) != hipSuccess) { g_gemm_cyc = nullptr; set_error("dev_gemm_stamps: no buffer"); return QEMB_ERR_ALLOC; } }
  HIP_TRY(hipMemsetAsync(g_gemm_cyc, 0, sizeof(long long) * g_gemm_cyc_cap, s));
  hipEvent_t t0, t1;
  HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
  g_gemm_cyc_blocks = 0; g_gemm_cyc_on = true;
  (void)hipEventRecord(t0, s);
  const int rc = dev_gemm_dispatch(d);
  (void)hipEventRecord(t1, s);
  (void)hipEventSynchronize(t1);
  g_gemm_cyc_on = false;
  float ms = 0.f; (void)hipEventElapsedTime(&ms, t0, t1);
  (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  if (rc) return rc;
  const size_t nw_ = (size_t)g_gemm_cyc_blocks * waves_per_wg;
  std::vector<long long> h(nw_ * 9);
  HIP_TRY(hipMemcpy(h.data(), g_gemm_cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  double sum[5] = {0, 0, 0, 0, 0};
  for (size_t w = 0; w < nw_; ++w) { for (int k = 0; k < 4; ++k) sum[k] += (double)h[w * 4 + k]; sum[4] += (double)h[nw_ * 4 + w]; }
  const double nw = (double)nw_;
  for (int k = 0; k < 3; ++k) out[k] = nw > 0 ? sum[k] / nw : 0.0;
  out[4] = nw > 0 ? sum[3] / nw : 0.0;      // entry -> end of main loop
  out[5] = nw > 0 ? sum[4] / nw : 0.0;      // entry -> exit
  out[6] = (double)g_gemm_cyc_blocks;
  for (int k = 0; k < 4; ++k) { double t = 0; for (size_t w = 0; w < nw_; ++w) t += (double)h[nw_ * 5 + w * 4 + k]; out[7 + k] = nw > 0 ? t / nw : 0.0; }
  out[3] = ms;
  return QEMB_OK;
}

// QEMB_GEMM_TRACE=1: every product goes through dev_gemm_probe and is logged -- a debugging aid, not a mode to run in
int dev_gemm(const GemmDesc& d) {
  static const bool trace = std::getenv("QEMB_GEMM_TRACE") != nullptr;
  if (!trace || dev_capturing()) return dev_gemm_dispatch(d);
  double ms = 0.0, ghz = 0.0; long long wg = 0;
  const int rc = dev_gemm_probe(d, &ms, &ghz, &wg);
  std::fprintf(stderr, "[qemb gemm] M=%lld N=%lld K=%lld batch=%lld a_kc=%d b_kc=%d cfg=%d beta=%g  %.4f ms  %.1f TF  wg-ticks/(256 CU x t) = %.2f GHz (%lld workgroups)\n",
               (long long)d.M, (long long)d.N, (long long)d.K, (long long)d.batch, (int)d.a_kcontig, (int)d.b_kcontig, d.cfg, d.beta, ms,
               ms > 0 ? 2.0 * d.M * d.N * d.K * d.batch / (ms * 1e9) : 0.0, ghz, wg);
  return rc;
}

static int dev_gemm_dispatch(const GemmDesc& d) {
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return QEMB_OK;
  if (d.K < 0 || !d.A || !d.B || !d.C) { set_error("dev_gemm: bad arguments"); return QEMB_ERR_ARG; }
  if (d.M > 0x3fffffff || d.N > 0x3fffffff || d.K > 0x3fffffff) {
    set_error("dev_gemm: dimension too large"); return QEMB_ERR_ARG;
  }
  hipStream_t s = hip_stream();
  g_gemm_flops += 2ll * d.M * d.N * d.K * d.batch;
  {  // QEMB_GEMM_SHAPELOG=<file>: one line per product in launch order (M N K batch a_kcontig b_kcontig cfg ksplit) -- no timing, no synchronisation; a single-stream
     // run under rocprofv3 --kernel-trace then pairs the i-th dgemm_mfma_kernel dispatch of the trace with the i-th line (tools/kernel_roofline.py: products by shape)
    static FILE* shapelog = [] { const char* e = std::getenv("QEMB_GEMM_SHAPELOG"); return (e && e[0]) ? std::fopen(e, "w") : (FILE*)nullptr; }();
    if (shapelog) {
      static std::mutex mu;
      std::lock_guard<std::mutex> lock(mu);
      std::fprintf(shapelog, "%lld %lld %lld %lld %d %d %d %d\n", (long long)d.M, (long long)d.N, (long long)d.K, (long long)d.batch, (int)d.a_kcontig, (int)d.b_kcontig, d.cfg, d.ksplit);
      std::fflush(shapelog);
    }
  }
  const bool vec2 = operand_vec2_ok(d.A, d.lda, d.strideA, d.a_kcontig ? d.K : d.M, d.a_kcontig != 0) &&
                    operand_vec2_ok(d.B, d.ldb, d.strideB, d.b_kcontig ? d.K : d.N, d.b_kcontig != 0);
  // tile choice: biggest tile that still gives the 256 CUs >= ~2 workgroups each; small problems
  // fall to 64x64 / 32x32 tiles so the grid is not a handful of blocks.
  const int64_t t128 = ((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch;
  const int64_t t64 = ((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch;
  // padding waste of the two main tilings (zero rows still occupy MFMA slots): M = o^2 = 400 wastes 22 % with
  // 128-row tiles but 11 % with 64-row tiles, and the smaller tile wins on the ladder shape (profiles/r01_gemm_*).
  const double w128 = (double)(((d.M + 127) / 128) * 128) * (double)(((d.N + 127) / 128) * 128);
  const double w64 = (double)(((d.M + 63) / 64) * 64) * (double)(((d.N + 63) / 64) * 64);
  int cfg;
  if (t128 >= 384) cfg = (w128 > 1.08 * w64) ? 1 : 0;
  else if (t64 >= 256) cfg = 1;
  else if (d.M >= 128 && d.N >= 128 && d.K >= 4096) cfg = 1;   // few tiles but a long K: split-K supplies the workgroups (200 x 200 x 80000: 0.24 vs 0.29 ms)
  else cfg = 2;
  // tall products with 193..224 columns: ONE 224-wide column tile (128 x 224) instead of two 128-wide ones, 12.5 % of which would be padding
  // (from ~1000 row tiles on: with fewer, two 128 x 128 workgroups per CU overlap their short-K prologues and epilogues better -- U = t2 . Lvv,
  //  625 row tiles, K = 200: 156 us against 163 us)
  if (vec2 && d.N > 192 && d.N <= 224 && d.M >= 128 * 1024) cfg = 34;
  if (d.cfg >= 0) cfg = d.cfg;
  if (t_gemm_force_cfg >= 0) cfg = t_gemm_force_cfg;
  // 3xx (s_memtime stamps) and 4xx-6xx (ablation: WRONG products by construction) exist for tools/ only: unreachable unless the process
  // says it is a measurement run
  if (cfg >= 300) {
    static const bool diag = std::getenv("QEMB_GEMM_DIAGNOSTICS") != nullptr;
    if (!diag) { set_error("dev_gemm: tile configs >= 300 are diagnostic instantiations (set QEMB_GEMM_DIAGNOSTICS=1 in a measurement run)"); return QEMB_ERR_ARG; }
  }
  // The large tiles run the MODE 1 main loop (explicit one-k-step-ahead LDS fragment reads, LDS stores spread behind the MFMA rows) when
  // the operands allow 16-byte loads; their scalar-load variants, the single-column wave tiles and the small / skinny tiles, which are
  // latency or HBM bound and want the two-tiles-deep register prefetch, keep the classic loop.
  switch (cfg) {
    case 0: return launch_layout<4, 4, 2, 2, 16, 0, 1>(d, s, vec2);   // 128 x 128, 4 waves
    case 1: return launch_layout<2, 2, 2, 2, 16, 0, 1>(d, s, vec2);   //  64 x  64, 4 waves
    case 2: return launch_layout<1, 1, 2, 2, 32>(d, s, vec2);   //  32 x  32, 4 waves
    case 4: return launch_layout<4, 4, 2, 4, 16, 0, 1>(d, s, vec2);   // 128 x 256, 8 waves
    case 10: return launch_layout<14, 1, 1, 8, 16>(d, s, vec2); // 224 x 128, 8 waves: all packed (i>=j) rows of o = 20 in ONE tile
    case 11: return launch_layout<7, 1, 1, 8, 16>(d, s, vec2);  // 112 x 128, 8 waves
    case 12: return launch_layout<4, 1, 1, 8, 16>(d, s, vec2);  //  64 x 128, 8 waves
    case 13: return launch_layout<7, 2, 2, 4, 16, 0, 1>(d, s, vec2);  // 224 x 128, 8 waves as 2 x 4: 9 LDS fragment reads per 14 MFMAs (15 for cfg 10)
    case 15: return launch_layout<6, 2, 2, 4, 16, 0, 1>(d, s, vec2);  // 192 x 128, 8 waves as 2 x 4 (the 190 antisymmetric pair rows of o = 20)
    case 33: return launch_layout<7, 2, 1, 4, 16, 0, 1>(d, s, vec2);  // 112 x 128, 4 waves, TWO workgroups per CU (66 KB of LDS each): short-K products
    case 34: return launch_layout<2, 7, 4, 2, 16, 0, 1>(d, s, vec2);  // 128 x 224, 8 waves as 4 x 2 (2 x 7 MFMA tiles per wave): tall products with 192 < N <= 224
    case 35: return launch_layout<5, 2, 2, 4, 16, 0, 1>(d, s, vec2);  // 160 x 128, 8 waves as 2 x 4 (5 x 2 MFMA tiles per wave): pair-row counts that 160 divides well (465 = npair(30))
    case 36: return launch_layout<5, 2, 1, 4, 16, 0, 1>(d, s, vec2);  //  80 x 128, 4 waves as 1 x 4 (5 x 2 per wave), two workgroups per CU: the 66-80 packed pair rows of n_occ = 12 (mid-size fragments, round 5)
    case 37: return launch_layout<3, 3, 2, 2, 16, 0, 1>(d, s, vec2);  //  96 x  96, 4 waves as 2 x 2 (3 x 3 per wave): square products of 1000-2000 rows and columns (the rings of mid-size fragments: 15 x 15 tiles at o v = 1440 fill 225 of 256 CUs in one round)
    case 38: return launch_layout<3, 2, 1, 4, 16, 0, 1>(d, s, vec2);  //  48 x 128, 4 waves as 1 x 4 (3 x 2 per wave): the 36-45 packed pair rows of n_occ = 9 (round 5: the 80-row tile spent 44 % of its MFMAs on padding there)
    case 236: return launch_layout<5, 2, 1, 4, 16>(d, s, vec2);
    case 237: return launch_layout<3, 3, 2, 2, 16>(d, s, vec2);
    case 20: return launch_layout<4, 1, 2, 2, 16>(d, s, vec2);   // 128 x  32, 4 waves: tall products with N = n_occ (the t1 contractions of ovvv)
    case 21: return launch_layout<1, 4, 2, 2, 16>(d, s, vec2);   //  32 x 128, 4 waves: the same with M = n_occ
    case 23: return launch_layout<7, 2, 2, 4, 16, 1, 1>(d, s, vec2);   // = 13 under its own kernel symbol (pp-ladder, + pairs)
    case 25: return launch_layout<6, 2, 2, 4, 16, 1, 1>(d, s, vec2);   // = 15 under its own kernel symbol (pp-ladder, - pairs)
    // the classic (MODE 0) main loop of EVERY tile that runs MODE 1 in production, cfg + 200: kept addressable for A/B measurements
    // (tools/gemm_modes.py) and for the bit-for-bit comparison tests/test_gpu_ops.py::test_gemm_mode1_equals_classic_loop runs, so that a
    // toolchain change that breaks the hand-counted LDS waits of MODE 1 is caught (same summation order: results must be identical)
    // diagnostic instantiations (TAG 2): per-wave s_memtime stamps around the per-tile barrier, read by qemb_op_gemm_stamps
    case 313: return launch_layout<7, 2, 2, 4, 16, 2, 1>(d, s, vec2);
    case 315: return launch_layout<6, 2, 2, 4, 16, 2, 1>(d, s, vec2);
    case 304: return launch_layout<4, 4, 2, 4, 16, 2, 1>(d, s, vec2);
    // ablation instantiations (WRONG results by construction; tools/gemm_ablation.py): 3 = no global loads in the main loop, 4 = half of the
    // A fragment reads, 5 = both -- what the clock the chip holds under the kernel owes to HBM / L2 traffic and to LDS reads
    case 413: return launch_layout<7, 2, 2, 4, 16, 3, 1>(d, s, vec2);
    case 513: return launch_layout<7, 2, 2, 4, 16, 4, 1>(d, s, vec2);
    case 613: return launch_layout<7, 2, 2, 4, 16, 5, 1>(d, s, vec2);
    case 404: return launch_layout<4, 4, 2, 4, 16, 3, 1>(d, s, vec2);
    case 504: return launch_layout<4, 4, 2, 4, 16, 4, 1>(d, s, vec2);
    case 604: return launch_layout<4, 4, 2, 4, 16, 5, 1>(d, s, vec2);
    case 200: return launch_layout<4, 4, 2, 2, 16>(d, s, vec2);
    case 201: return launch_layout<2, 2, 2, 2, 16>(d, s, vec2);
    case 204: return launch_layout<4, 4, 2, 4, 16>(d, s, vec2);
    case 213: return launch_layout<7, 2, 2, 4, 16>(d, s, vec2);
    case 215: return launch_layout<6, 2, 2, 4, 16>(d, s, vec2);
    case 233: return launch_layout<7, 2, 1, 4, 16>(d, s, vec2);
    case 234: return launch_layout<2, 7, 4, 2, 16>(d, s, vec2);
    case 235: return launch_layout<5, 2, 2, 4, 16>(d, s, vec2);
    default: set_error("dev_gemm: unknown tile config"); return QEMB_ERR_ARG;
  }
}

// grouped execution of several fragments' GEMMs in one launch (dev_ops_hip.hip "grouped launches"): registered there
// (the tiles small fragments run on: 32 x 32, 64 x 64 and the skinny 128 x 32 / 32 x 128, every operand layout, 16- and 8-byte loads)
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int MODE>
static void register_gemm_tile() {
#define QEMB_REG(AK, BKC)                                                                                                                          \
  register_groupable<dgemm_mfma_body<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 2, 0, MODE>, WAVES_M * WAVES_N * 64, GemmKArgs>(                        \
      (const void*)dgemm_mfma_kernel<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 2, 0, MODE>);                                                           \
  register_groupable<dgemm_mfma_body<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 1, 0, 0>, WAVES_M * WAVES_N * 64, GemmKArgs>(                           \
      (const void*)dgemm_mfma_kernel<WM, WN, WAVES_M, WAVES_N, BK, AK, BKC, 1, 0, 0>);
  QEMB_REG(true, true) QEMB_REG(true, false) QEMB_REG(false, true) QEMB_REG(false, false)
#undef QEMB_REG
}
void register_groupable_gemm() {
  register_gemm_tile<1, 1, 2, 2, 32, 0>();     // cfg 2
  register_gemm_tile<2, 2, 2, 2, 16, 1>();     // cfg 1
  register_gemm_tile<4, 1, 2, 2, 16, 0>();     // cfg 20
  register_gemm_tile<1, 4, 2, 2, 16, 0>();     // cfg 21
  register_groupable<splitk_reduce_kernel_body, 256, const double*, int, long long, long long, double*, long long, long long, double, double>((const void*)splitk_reduce_kernel);
  register_groupable<splitk_reduce_wave_kernel_body, 256, const double*, int, long long, long long, double*, long long, long long, double, double>((const void*)splitk_reduce_wave_kernel);
}

}  // namespace qemb
