// linalg_f64.hip -- wavefront Jacobi eigen / SVD and small dense factorizations for gfx950.
//
// Schmidt decomposition (reference: molbe/pfrag.py:403-494 `eigh(Denv)`; kbe/solver.py:9-46 `svd`) and the
// Fock diagonalisations of the fragment RHF (molbe/helper.py:73-151) are eigenproblems with no GEMM shape:
// they run as one-sided (Hestenes) Jacobi sweeps.  A "round" applies n/2 independent plane rotations
// (round-robin tournament ordering), one workgroup per vector pair; within a workgroup every wave streams
// its slice of the two vectors (coalesced), the three Gram entries are reduced across the wave64s, and the
// rotated vectors are written back.  No MFMA -- the work is dot products and axpys out of L2.
//
// Symmetric eigenproblems are solved as the SVD of the shifted matrix A + sigma*I (sigma from Gershgorin,
// so the shifted matrix is positive definite and singular vectors == eigenvectors even when A has +/- pairs).
//
// Also here: blocked Cholesky and triangular inverse for the DF metric (reference: eri_onthefly.py:108,141;
// _cpp/eri_sparse_DF.cpp:611-621), built from a 32x32 LDS diagonal-block kernel plus the MFMA GEMM.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dev_ops.h"
#include "hip_common.h"

namespace qemb {

// pooled device allocations (dev_alloc parks freed blocks, so the per-call scratch of the Jacobi / Cholesky drivers
// costs no hipMalloc / hipFree after the first call)
#define QTRY_ALLOC(ptr, bytes)                                                                  \
  do { void* _q = nullptr; int _rc = dev_alloc(&_q, (bytes)); if (_rc) return _rc; ptr = (decltype(ptr))_q; } while (0)

// ------------------------------------------------------------------------------------------------
// Jacobi core: orthogonalise the rows of W (nvec x len, ld = ldw), accumulate rotations in Vt (nvec x nvec)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// round-robin pairing (circle method): np players (even), round r in [0, np-1), slot k in [0, np/2)
__device__ __forceinline__ void rr_pair(int np, int r, int k, int& p, int& q) {
  const int m = np - 1;
  if (k == 0) { p = m; q = r; }
  else { p = (r + k) % m; q = (r - k + m) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

__global__ void __launch_bounds__(256) jacobi_round_kernel(double* W, long long ldw, long long len, double* Vt, int nvec, int np,
                                                           int round, double tol, double floor2, unsigned long long* offmax_bits) {
  __shared__ double sh[3][4];
  __shared__ double cs[2];
  int p, q;
  rr_pair(np, round, blockIdx.x, p, q);
  if (q >= nvec) return;  // phantom player of an odd tournament
  double* wp = W + (long long)p * ldw;
  double* wq = W + (long long)q * ldw;
  double a = 0.0, b = 0.0, g = 0.0;
  for (long long i = threadIdx.x; i < len; i += blockDim.x) {
    const double x = wp[i], y = wq[i];
    a += x * x; b += y * y; g += x * y;
  }
  a = wsum(a); b = wsum(b); g = wsum(g);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = a; sh[1][w] = b; sh[2][w] = g; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double A = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    const double B = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    const double G = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    double c = 1.0, s = 0.0;
    if (A > floor2 && B > floor2) {
      const double rel = fabs(G) / sqrt(A * B);
      if (rel > tol) {
        const double zeta = (B - A) / (2.0 * G);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        c = 1.0 / sqrt(1.0 + t * t);
        s = c * t;
        atomicMax(offmax_bits, (unsigned long long)__double_as_longlong(rel));
      }
    }
    cs[0] = c; cs[1] = s;
  }
  __syncthreads();
  const double c = cs[0], s = cs[1];
  if (s == 0.0) return;
  for (long long i = threadIdx.x; i < len; i += blockDim.x) {
    const double x = wp[i], y = wq[i];
    wp[i] = c * x - s * y;
    wq[i] = s * x + c * y;
  }
  if (Vt) {
    double* vp = Vt + (long long)p * nvec;
    double* vq = Vt + (long long)q * nvec;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
      const double x = vp[i], y = vq[i];
      vp[i] = c * x - s * y;
      vq[i] = s * x + c * y;
    }
  }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double lin_dpp_f64(double x) {
  const unsigned long long u = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_allsum(double v) {      // the sum over the 64 lanes, in every lane (wave-uniform), no LDS crossbar
  v += lin_dpp_f64<0xB1, 0xF>(v);
  v += lin_dpp_f64<0x4E, 0xF>(v);
  v += lin_dpp_f64<0x141, 0xF>(v);
  v += lin_dpp_f64<0x140, 0xF>(v);
  v += lin_dpp_f64<0x142, 0xA>(v);
  v += lin_dpp_f64<0x143, 0xC>(v);
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// plane rotation that annihilates g = <x,y> given a = <x,x>, b = <y,y>: the inner rotation (|theta| <= pi/4) in its half-angle form
//   r = sqrt(d^2 + h^2), d = b - a, h = 2 g;  cos^2 = (1 + |d|/r)/2;  sin = sign(d) h / (2 r cos)
// -- two reciprocal square roots instead of the three divisions and three square roots of the tangent form (the scalar arithmetic of a
// rotation is serial work every lane of the wave waits for); same angle, cos^2 + sin^2 = 1 to rounding.
__device__ __forceinline__ void jacobi_cs(double a, double b, double g, double& c, double& sn) {
  const double d = b - a, h = 2.0 * g;
  const double rinv = rsqrt(d * d + h * h);
  const double c2 = 0.5 + 0.5 * fabs(d) * rinv;
  const double cinv = rsqrt(c2);
  c = c2 * cinv;
  sn = (d >= 0.0 ? 0.5 : -0.5) * h * rinv * cinv;
}
// Single-workgroup variant for small problems (nvec, len <= JS_MAX): W and Vt live in LDS, every round is one pass of
// the workgroup's 16 waves over the n/2 pairs (one pair per wave at a time), rounds and sweeps are separated by
// __syncthreads instead of kernel launches.  A 41-orbital fragment Fock matrix (octane BE2) needs ~330 rounds per
// eigh: one launch here instead of 330.
constexpr int JS_MAX = 96;
__global__ void __launch_bounds__(1024) jacobi_small_kernel(double* Wg, long long ldw, int len, double* Vtg, int nvec, int np,
                                                            double tol, double floor2, int max_sweeps, int* sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ldW = len + 1, ldV = nvec + 1;
  double* W = lds;
  double* V = lds + (size_t)nvec * ldW;
  __shared__ int rotated;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  for (int t = tid; t < nvec * len; t += blockDim.x) W[(t / len) * ldW + (t % len)] = Wg[(long long)(t / len) * ldw + (t % len)];
  for (int t = tid; t < nvec * nvec; t += blockDim.x) V[(t / nvec) * ldV + (t % nvec)] = Vtg[(long long)(t / nvec) * nvec + (t % nvec)];
  __syncthreads();
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (int r = 0; r < np - 1; ++r) {
      for (int k = wave; k < np / 2; k += nwaves) {
        int p, q;
        rr_pair(np, r, k, p, q);
        if (q >= nvec) continue;
        double* wp = W + p * ldW; double* wq = W + q * ldW;
        double a = 0.0, b = 0.0, g = 0.0;
        for (int i = lane; i < len; i += 64) { const double x = wp[i], y = wq[i]; a += x * x; b += y * y; g += x * y; }
        a = wave_allsum(a); b = wave_allsum(b); g = wave_allsum(g);      // DPP: no LDS crossbar (round 3; 18 ds_bpermute per pair before)
        double c = 1.0, sn = 0.0;
        if (a > floor2 && b > floor2 && g * g > tol * tol * a * b) jacobi_cs(a, b, g, c, sn);      // half-angle form: two rsqrt (round 3)
        if (sn != 0.0) {
          if (lane == 0) rotated = 1;
          for (int i = lane; i < len; i += 64) { const double x = wp[i], y = wq[i]; wp[i] = c * x - sn * y; wq[i] = sn * x + c * y; }
          double* vp = V + p * ldV; double* vq = V + q * ldV;
          for (int i = lane; i < nvec; i += 64) { const double x = vp[i], y = vq[i]; vp[i] = c * x - sn * y; vq[i] = sn * x + c * y; }
        }
      }
      __syncthreads();
    }
    const int any = rotated;
    __syncthreads();
    if (!any) { ++sweep; break; }
  }
  for (int t = tid; t < nvec * len; t += blockDim.x) Wg[(long long)(t / len) * ldw + (t % len)] = W[(t / len) * ldW + (t % len)];
  for (int t = tid; t < nvec * nvec; t += blockDim.x) Vtg[(long long)(t / nvec) * nvec + (t % nvec)] = V[(t / nvec) * ldV + (t % nvec)];
  if (tid == 0) sweeps_out[0] = (sweep <= max_sweeps && !rotated) ? sweep : -1;
}

// ------------------------------------------------------------------------------------------------
// Symmetric eigenproblem of a SMALL matrix (n <= JE_MAX) in ONE launch (round 5): classical two-sided cyclic Jacobi, A <- J^T A J, with A and V
// in the LDS of one workgroup.  The one-sided kernel above needs three length-n dot products and two wave-wide reductions per pair before it
// can rotate, and a wave per pair: 3.4 us per round at n = 42, 0.7-1.2 ms per eigensolve of a fragment Fock matrix -- a quarter of an octane BE2
// sweep went into it.  Two-sided, the angle of a pair comes from three matrix ELEMENTS (a_pp, a_qq, a_pq; untouched by the other pairs of the
// round), so a round is: (0) one lane per pair computes (c, s); (1) every thread applies the row rotations, all n^2 elements in parallel; (2) the
// column rotations of A and V -- three workgroup barriers, no reductions.  Same rotation angle as the Hestenes form (jacobi_cs on the 2 x 2 block),
// same round-robin order of pairs, a rotation only where |a_pq| > tol x (largest absolute row sum); a sweep whose largest rotated element is below
// stop_below x that scale is the last (quadratic convergence: what is left is its square).  Eigenvalues are ranked and the eigenvector columns
// permuted inside the kernel (ascending, ties in index order), so the whole eigensolve is one launch and one status word back: before it was
// seven launches and four stream waits (Gershgorin shift, identity, sweeps, Rayleigh quotients, host sort, gather).
constexpr int JE_MAX = 96;
__global__ void __launch_bounds__(1024) jacobi_eigh_small_kernel(const double* __restrict__ Ag, int n, double* __restrict__ w_out, double* __restrict__ V_out,
                                                                 double tol, double stop_below, int max_sweeps, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ld = n + 1 + (n & 1);                  // odd row stride: column walks are conflict free
  const int np = n + (n & 1), nh = np >> 1;
  double* A = lds;
  double* V = A + (size_t)n * ld;
  double* pc = V + (size_t)n * ld;                 // per pair of the round: cos, sin
  double* ps = pc + nh;
  double* rsum = ps + nh;                          // n row sums (scale), later the eigenvalues
  int* pp = (int*)(rsum + n);                      // per pair: p, q (q < 0: no rotation)
  int* pq = pp + nh;
  int* rank = pq + nh;
  __shared__ unsigned long long offbits;
  __shared__ double scale_sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; A[i * ld + j] = Ag[t]; V[i * ld + j] = (i == j) ? 1.0 : 0.0; }
  if (tid == 0) offbits = 0ull;
  __syncthreads();
  for (int i = tid; i < n; i += nt) { double a = 0.0; for (int j = 0; j < n; ++j) a += fabs(A[i * ld + j]); rsum[i] = a; }
  __syncthreads();
  if (tid == 0) { double m = 0.0; for (int i = 0; i < n; ++i) m = fmax(m, rsum[i]); scale_sh = m; }
  __syncthreads();
  const double scale = scale_sh, thr = tol * scale;
  int sweep = 0;
  bool done = (scale == 0.0);
  for (; sweep < max_sweeps && !done; ++sweep) {
    for (int r = 0; r < np - 1; ++r) {
      if (tid < nh) {
        int p, q;
        rr_pair(np, r, tid, p, q);
        double c = 1.0, sn = 0.0;
        int qq = -1;
        if (q < n) {
          const double g = A[p * ld + q];
          if (fabs(g) > thr) {
            jacobi_cs(A[p * ld + p], A[q * ld + q], g, c, sn);
            qq = q;
            atomicMax(&offbits, (unsigned long long)__double_as_longlong(fabs(g)));      // (non-negative doubles order like their bit patterns)
          }
        }
        pc[tid] = c; ps[tid] = sn; pp[tid] = p; pq[tid] = qq;
      }
      __syncthreads();
      for (int t = tid; t < nh * n; t += nt) {       // rows p, q of A
        const int k = t / n, j = t - k * n, q = pq[k];
        if (q < 0) continue;
        const int p = pp[k];
        const double c = pc[k], sn = ps[k];
        const double x = A[p * ld + j], y = A[q * ld + j];
        A[p * ld + j] = c * x - sn * y; A[q * ld + j] = sn * x + c * y;
      }
      __syncthreads();
      for (int t = tid; t < nh * n; t += nt) {       // columns p, q of A and of V
        const int k = t / n, i = t - k * n, q = pq[k];
        if (q < 0) continue;
        const int p = pp[k];
        const double c = pc[k], sn = ps[k];
        double x = A[i * ld + p], y = A[i * ld + q];
        double xn = c * x - sn * y, yn = sn * x + c * y;
        if (i == p) yn = 0.0;                        // the annihilated element, exactly
        if (i == q) xn = 0.0;
        A[i * ld + p] = xn; A[i * ld + q] = yn;
        x = V[i * ld + p]; y = V[i * ld + q];
        V[i * ld + p] = c * x - sn * y; V[i * ld + q] = sn * x + c * y;
      }
      __syncthreads();
    }
    const double offmax = __longlong_as_double((long long)offbits);
    __syncthreads();
    if (tid == 0) offbits = 0ull;
    __syncthreads();
    if (offmax == 0.0 || offmax < stop_below * scale) { done = true; }
  }
  // eigenvalues ascending (ties in index order), eigenvector columns permuted accordingly
  for (int i = tid; i < n; i += nt) rsum[i] = A[i * ld + i];
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const double wi = rsum[i];
    int rk = 0;
    for (int j = 0; j < n; ++j) { const double wj = rsum[j]; rk += (wj < wi || (wj == wi && j < i)) ? 1 : 0; }
    rank[i] = rk;
    w_out[rk] = wi;
  }
  __syncthreads();
  for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; V_out[i * n + rank[j]] = V[i * ld + j]; }
  if (tid == 0) status[0] = done ? sweep : -1;
}

// The same sweeps with ONE barrier per round (n <= JE_DB_MAX): A is double buffered, and the thread that owns the element pair (i; p, q) of column pair
// k = (p, q) forms dst[i][p], dst[i][q] = (J^T src J)[i][p], [i][q] in one go -- it works out both rotations it needs itself (its column pair's and the one
// row i takes part in: six elements of src and two jacobi_cs, the same numbers in every thread that needs them) -- and rotates its two elements of V in
// place.  The partner of index i in round r of the round-robin is (2 r - i) mod (np - 1) (np - 1 itself pairs with r).  Largest rotated element per
// thread in a register, combined once per sweep.  Same pairs, same angles, same order of rounds as jacobi_eigh_small_kernel.
constexpr int JE_DB_MAX = 80;
// Fused form for the fragment RHF (round 5; Cp != nullptr or dm_out != nullptr): the matrix is first rotated into the basis Cp (A = Cp^T F Cp, the orbitals of the last
// SCF cycle, in which F is nearly diagonal), the eigenvectors are rotated back (C = Cp V, columns in ascending order of the eigenvalues), written to V_out and --
// when asked -- a second time to C2_out (which may be Cp itself: the next cycle's basis), and dm_out = 2 C_occ C_occ^T of the lowest `nocc` columns.  One launch
// where the SCF cycle of a small fragment made seven (two products in, the eigensolve, one product out, a copy, the density product).
__global__ void __launch_bounds__(1024) jacobi_eigh_small_db_kernel(const double* __restrict__ Ag, int n, double* __restrict__ w_out, double* V_out,
                                                                    double tol, double stop_below, int max_sweeps, int* __restrict__ status,
                                                                    const double* Cp, double* C2_out, int nocc, double* __restrict__ dm_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ld = n + 1 + (n & 1);
  const int np = n + (n & 1), nh = np >> 1, m = np - 1;
  double* A0 = lds;
  double* A1 = A0 + (size_t)n * ld;
  double* V = A1 + (size_t)n * ld;
  double* rsum = V + (size_t)n * ld;
  int* rank = (int*)(rsum + n);
  __shared__ unsigned long long offbits;
  __shared__ double scale_sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (Cp) {
    // A0 = Cp^T (F Cp): F in A1, Cp in V, the half product in A0, the result back in A1 -- then A1 moves to A0 and V becomes the identity
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; A1[i * ld + j] = Ag[t]; V[i * ld + j] = Cp[t]; }
    __syncthreads();
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; double a = 0.0; for (int k = 0; k < n; ++k) a += A1[i * ld + k] * V[k * ld + j]; A0[i * ld + j] = a; }
    __syncthreads();
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; double a = 0.0; for (int k = 0; k < n; ++k) a += V[k * ld + i] * A0[k * ld + j]; A1[i * ld + j] = a; }
    __syncthreads();
    // (the product is symmetric up to rounding; the sweeps read the upper triangle for the angles and keep both halves: make it exactly symmetric)
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; A0[i * ld + j] = (i <= j) ? A1[i * ld + j] : A1[j * ld + i]; V[i * ld + j] = (i == j) ? 1.0 : 0.0; }
  } else {
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; A0[i * ld + j] = Ag[t]; V[i * ld + j] = (i == j) ? 1.0 : 0.0; }
  }
  if (tid == 0) offbits = 0ull;
  __syncthreads();
  for (int i = tid; i < n; i += nt) { double a = 0.0; for (int j = 0; j < n; ++j) a += fabs(A0[i * ld + j]); rsum[i] = a; }
  __syncthreads();
  if (tid == 0) { double mx = 0.0; for (int i = 0; i < n; ++i) mx = fmax(mx, rsum[i]); scale_sh = mx; }
  __syncthreads();
  const double scale = scale_sh, thr = tol * scale;
  // this thread's items (i, k), fixed for the whole solve: t = k * n + i for t = tid, tid + nt, ...
  constexpr int MAXI = 4;
  int it_i[MAXI], it_k[MAXI];
  const int nitems = nh * n;
#pragma unroll
  for (int a = 0; a < MAXI; ++a) { const int t = tid + a * nt; it_k[a] = (t < nitems) ? t / n : -1; it_i[a] = (t < nitems) ? t - (t / n) * n : 0; }
  double* src = A0; double* dst = A1;
  int sweep = 0;
  bool done = (scale == 0.0);
  for (; sweep < max_sweeps && !done; ++sweep) {
    double off_seen = 0.0;
    for (int r = 0; r < m; ++r) {
#pragma unroll
      for (int a = 0; a < MAXI; ++a) {
        const int k = it_k[a];
        if (k < 0) continue;
        const int i = it_i[a];
        // column pair k of round r
        int p, q;
        if (k == 0) { p = m; q = r; } else { p = r + k; if (p >= m) p -= m; q = r - k; if (q < 0) q += m; }
        if (p > q) { const int t = p; p = q; q = t; }
        double ck = 1.0, sk = 0.0;
        const bool colpair = q < n;
        if (colpair) {
          const double g = src[p * ld + q];
          if (fabs(g) > thr) { jacobi_cs(src[p * ld + p], src[q * ld + q], g, ck, sk); off_seen = fmax(off_seen, fabs(g)); }
        }
        // the pair row i belongs to
        int ip = (i == m) ? r : ((i == r) ? m : 2 * r - i);
        if (i != m && i != r) { if (ip < 0) ip += m; else if (ip >= m) ip -= m; }
        double ci = 1.0, si = 0.0;
        if (ip < n) {
          const int lo = i < ip ? i : ip, hi = i < ip ? ip : i;
          const double g = src[lo * ld + hi];
          if (fabs(g) > thr) {
            double c, sn;
            jacobi_cs(src[lo * ld + lo], src[hi * ld + hi], g, c, sn);
            ci = c; si = (i == lo) ? -sn : sn;          // row_lo' = c row_lo - s row_hi,  row_hi' = s row_lo + c row_hi
          }
        } else ip = i;
        const double up = ci * src[i * ld + p] + si * src[ip * ld + p];
        if (colpair) {
          const double uq = ci * src[i * ld + q] + si * src[ip * ld + q];
          double xn = ck * up - sk * uq, yn = sk * up + ck * uq;
          if (sk != 0.0) { if (i == p) yn = 0.0; if (i == q) xn = 0.0; }
          dst[i * ld + p] = xn; dst[i * ld + q] = yn;
          if (sk != 0.0) { const double x = V[i * ld + p], y = V[i * ld + q]; V[i * ld + p] = ck * x - sk * y; V[i * ld + q] = sk * x + ck * y; }
        } else {
          dst[i * ld + p] = up;                          // column p sits this round out (odd n): only its row rotation
        }
      }
      __syncthreads();
      double* t = src; src = dst; dst = t;
    }
    if (off_seen > 0.0) atomicMax(&offbits, (unsigned long long)__double_as_longlong(off_seen));
    __syncthreads();
    const double offmax = __longlong_as_double((long long)offbits);
    __syncthreads();
    if (tid == 0) offbits = 0ull;
    __syncthreads();
    if (offmax == 0.0 || offmax < stop_below * scale) done = true;
  }
  for (int i = tid; i < n; i += nt) rsum[i] = src[i * ld + i];
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const double wi = rsum[i];
    int rk = 0;
    for (int j = 0; j < n; ++j) { const double wj = rsum[j]; rk += (wj < wi || (wj == wi && j < i)) ? 1 : 0; }
    rank[i] = rk;
    w_out[rk] = wi;
  }
  __syncthreads();
  if (!Cp && !dm_out && !C2_out) {
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, j = t - i * n; V_out[i * n + rank[j]] = V[i * ld + j]; }
  } else {
    // C = Cp V (or V itself), columns ranked, staged in the buffer the sweeps no longer need -- C2_out may be Cp, which every thread still reads here
    double* Cs = dst;
    for (int t = tid; t < n * n; t += nt) {
      const int i = t / n, j = t - i * n;
      double a;
      if (Cp) { a = 0.0; for (int k = 0; k < n; ++k) a += Cp[i * n + k] * V[k * ld + j]; }
      else a = V[i * ld + j];
      Cs[i * ld + rank[j]] = a;
    }
    __syncthreads();
    for (int t = tid; t < n * n; t += nt) {
      const int i = t / n, j = t - i * n;
      const double c = Cs[i * ld + j];
      V_out[t] = c;
      if (C2_out) C2_out[t] = c;
      if (dm_out) { double a = 0.0; for (int k = 0; k < nocc; ++k) a += Cs[i * ld + k] * Cs[j * ld + k]; dm_out[t] = 2.0 * a; }
    }
  }
  if (tid == 0) status[0] = done ? sweep : -1;
}

// F = h + J - K/2, err = F D - D F, scal[0] = sum (h + F) o D (twice the SCF energy), scal[1] = sum err^2: the bookkeeping of one SCF cycle of a SMALL fragment
// (n <= JE_MAX; F and D in the LDS of one workgroup) in one launch -- it was eight (the Fock combination, h + F, two reductions of two launches each, two n^3 products).
// The two sums are formed in a fixed order (per thread over its elements, then a tree over the threads): the same numbers on every run.
__global__ void __launch_bounds__(1024) scf_fock_small_kernel(int n, const double* __restrict__ h, const double* __restrict__ J, const double* __restrict__ K,
                                                              const double* __restrict__ D, double* __restrict__ F_out, double* __restrict__ err_out,
                                                              double* __restrict__ scal) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ld = n + 1 + (n & 1);
  double* F = lds;
  double* Dm = F + (size_t)n * ld;
  double* red = Dm + (size_t)n * ld;      // 2 x blockDim.x
  const int tid = threadIdx.x, nt = blockDim.x;
  double e2 = 0.0, g2 = 0.0;
  for (int t = tid; t < n * n; t += nt) {
    const int i = t / n, j = t - i * n;
    const double hh = h[t], f = hh + J[t] - 0.5 * K[t], d = D[t];
    F[i * ld + j] = f; Dm[i * ld + j] = d; F_out[t] = f;
    e2 += (hh + f) * d;
  }
  __syncthreads();
  for (int t = tid; t < n * n; t += nt) {
    const int i = t / n, j = t - i * n;
    double a = 0.0;
    for (int k = 0; k < n; ++k) a += F[i * ld + k] * Dm[k * ld + j] - Dm[i * ld + k] * F[k * ld + j];
    err_out[t] = a;
    g2 += a * a;
  }
  red[tid] = e2; red[nt + tid] = g2;
  __syncthreads();
  for (int s = nt >> 1; s > 0; s >>= 1) {
    if (tid < s) { red[tid] += red[tid + s]; red[nt + tid] += red[nt + tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { scal[0] = red[0]; scal[1] = red[nt]; }
}
// Dp[P(r,s)] = D[r,s] + D[s,r] (r > s), D[r,r]: the packed density of the Coulomb pass over the 4-fold packed block (four launches before: transpose, sum, diagonal, pack)
__global__ void __launch_bounds__(256) pack_density_sym_kernel(int n, const double* __restrict__ D, double* __restrict__ Dp) {
  const long long np = (long long)n * (n + 1) / 2;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (long long)gridDim.x * blockDim.x) {
    // row r of the lower triangle that holds p: r (r + 1) / 2 <= p
    long long r = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > p) --r;
    while ((r + 1) * (r + 2) / 2 <= p) ++r;
    const long long c = p - r * (r + 1) / 2;
    Dp[p] = (r == c) ? D[r * n + r] : D[r * n + c] + D[c * n + r];
  }
}

// ------------------------------------------------------------------------------------------------
// Block rounds (round 3): the same Hestenes rotations, grouped so that a workgroup keeps TWO blocks of B vectors (W rows and their Vt rows)
// in LDS and applies every rotation between them -- B inner rounds of B disjoint pairs, one pair per wave, separated by workgroup barriers
// -- before the vectors go back to memory.  A sweep is then nb - 1 launches (nb = ceil(nvec / B) blocks in a round-robin tournament) instead
// of nvec - 1: 14 instead of 219 at nvec = 220, B = 16, and each launch does 16 (31 in the first round, which also rotates the pairs inside
// each block) rounds of work out of LDS instead of one out of L2.  Same pairs per sweep, every pair exactly once; only the order differs.
// (Used for 96 < nvec and 2 B (len + nvec) doubles fitting the LDS of a CU; the per-pair kernel above remains for longer vectors.)
// B = vectors per block = waves per workgroup; NPL = ceil((len + nvec) / 64) register slots per lane for one [W row | Vt row]
template <int B, int NPL>
__global__ void __launch_bounds__(B * 64) jacobi_block_round_kernel(double* __restrict__ W, long long ldw, int len, double* __restrict__ Vt, int nvec, int nb,
                                                                    int nbp, int round, double tol, double floor2,
                                                                    unsigned long long* __restrict__ offmax_bits) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double wg_off[B];
  int P, Q;
  rr_pair(nbp, round, blockIdx.x, P, Q);      // P < Q; Q == nb: the phantom block of an odd tournament
  const bool haveQ = Q < nb;
  const bool do_diag = (round == 0);          // every block plays exactly once in round 0: rotate the pairs INSIDE the blocks there
  if (!haveQ && !do_diag) return;
  const int ldr = len + nvec;                 // one row: the W row followed by its Vt row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowsP = min(B, nvec - P * B), rowsQ = haveQ ? min(B, nvec - Q * B) : 0;
  const double tol2 = tol * tol;
  double off_seen = 0.0;
  // LDS rows 0 .. B-1: block Q (the rows that travel between waves); rows B .. 2B-1: block P, only while round 0 rotates inside the blocks
  auto load_row = [&](int gr, double* dst) {
    const double* src = W + (long long)gr * ldw;
    for (int i = lane; i < len; i += 64) dst[i] = src[i];
    const double* vs = Vt + (long long)gr * nvec;
    for (int i = lane; i < nvec; i += 64) dst[len + i] = vs[i];
  };
  auto store_row = [&](int gr, const double* src) {
    double* dst = W + (long long)gr * ldw;
    for (int i = lane; i < len; i += 64) dst[i] = src[i];
    double* vd = Vt + (long long)gr * nvec;
    for (int i = lane; i < nvec; i += 64) vd[i] = src[len + i];
  };
  if (wave < rowsQ) load_row(Q * B + wave, lds + (size_t)wave * ldr);
  double pr[NPL];                             // this wave's P row, in registers for the whole launch
  if (do_diag) {
    if (wave < rowsP) load_row(P * B + wave, lds + (size_t)(B + wave) * ldr);
    __syncthreads();
    // pairs inside block Q (waves 0 .. B/2-1, LDS rows 0..) and inside block P (waves B/2 .. B-1, LDS rows B..): B - 1 round-robin rounds
    const int h = wave / (B / 2), k = wave % (B / 2), nr = h == 0 ? rowsQ : rowsP;
    for (int r = 0; r < B - 1; ++r) {
      int p, q;
      rr_pair(B, r, k, p, q);
      if (q < nr) {
        double* wp = lds + (size_t)(h * B + p) * ldr; double* wq = lds + (size_t)(h * B + q) * ldr;
        double a = 0.0, b = 0.0, g = 0.0;
        for (int i = lane; i < len; i += 64) { const double x = wp[i], y = wq[i]; a += x * x; b += y * y; g += x * y; }
        a = wave_allsum(a); b = wave_allsum(b); g = wave_allsum(g);
        if (a > floor2 && b > floor2 && g * g > tol2 * a * b) {
          off_seen = fmax(off_seen, fabs(g) * rsqrt(a * b));
          double c, sn; jacobi_cs(a, b, g, c, sn);
          for (int i = lane; i < ldr; i += 64) { const double x = wp[i], y = wq[i]; wp[i] = c * x - sn * y; wq[i] = sn * x + c * y; }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k2 = 0; k2 < NPL; ++k2) { const int i = lane + 64 * k2; pr[k2] = (wave < rowsP && i < ldr) ? lds[(size_t)(B + wave) * ldr + i] : 0.0; }
  } else {
    const double* src = W + (long long)(P * B + (wave < rowsP ? wave : 0)) * ldw;
    const double* vs = Vt + (long long)(P * B + (wave < rowsP ? wave : 0)) * nvec;
#pragma unroll
    for (int k2 = 0; k2 < NPL; ++k2) {
      const int i = lane + 64 * k2;
      pr[k2] = (wave < rowsP && i < ldr) ? (i < len ? src[i] : vs[i - len]) : 0.0;
    }
    __syncthreads();
  }
  if (haveQ) {
    // pairs between the blocks: wave w keeps P row w and meets Q row (w + shift) % B in inner round `shift`; only the Q rows pass through LDS
    double a = 0.0;                           // <p,p> of this wave's row: carried along (a' = c^2 a - 2 c s g + s^2 b after a rotation), recomputed every launch
#pragma unroll
    for (int k2 = 0; k2 < NPL; ++k2) { const int i = lane + 64 * k2; a += (i < len) ? pr[k2] * pr[k2] : 0.0; }
    a = wave_allsum(a);
    for (int shift = 0; shift < B; ++shift) {
      const int qa = (wave + shift) % B;
      if (wave < rowsP && qa < rowsQ) {
        double* wq = lds + (size_t)qa * ldr;
        double qr[NPL];
        double b = 0.0, g = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < NPL; ++k2) {
          const int i = lane + 64 * k2;
          qr[k2] = (i < ldr) ? wq[i] : 0.0;
          if (i < len) { b += qr[k2] * qr[k2]; g += pr[k2] * qr[k2]; }
        }
        b = wave_allsum(b); g = wave_allsum(g);
        if (a > floor2 && b > floor2 && g * g > tol2 * a * b) {
          off_seen = fmax(off_seen, fabs(g) * rsqrt(a * b));
          double c, sn; jacobi_cs(a, b, g, c, sn);
#pragma unroll
          for (int k2 = 0; k2 < NPL; ++k2) {
            const int i = lane + 64 * k2;
            const double x = pr[k2], y = qr[k2];
            pr[k2] = c * x - sn * y;
            if (i < ldr) wq[i] = sn * x + c * y;
          }
          a = 0.0;                            // (recomputed from the rotated row: no drift of the carried norm)
#pragma unroll
          for (int k2 = 0; k2 < NPL; ++k2) { const int i = lane + 64 * k2; a += (i < len) ? pr[k2] * pr[k2] : 0.0; }
          a = wave_allsum(a);
        }
      }
      __syncthreads();
    }
  }
  if (wave < rowsQ) store_row(Q * B + wave, lds + (size_t)wave * ldr);
  if (wave < rowsP) {
    double* dst = W + (long long)(P * B + wave) * ldw;
    double* vd = Vt + (long long)(P * B + wave) * nvec;
#pragma unroll
    for (int k2 = 0; k2 < NPL; ++k2) {
      const int i = lane + 64 * k2;
      if (i < len) dst[i] = pr[k2]; else if (i < ldr) vd[i - len] = pr[k2];
    }
  }
  if (lane == 0) wg_off[wave] = off_seen;
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int w = 0; w < B; ++w) m = fmax(m, wg_off[w]);
    if (m > 0.0) atomicMax(offmax_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// out[i] = sum_k X[i*ldx+k] * Y[i*ldy+k]
__global__ void __launch_bounds__(256) rowdot_kernel(int nrow, long long len, const double* X, long long ldx, const double* Y, long long ldy, double* out) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double acc = 0.0;
  for (long long k = threadIdx.x; k < len; k += blockDim.x) acc += X[i * ldx + k] * Y[i * ldy + k];
  acc = wsum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[i] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[i] = sum_k |X[i*ld+k]|
__global__ void __launch_bounds__(256) rowabssum_kernel(int nrow, long long len, const double* X, long long ld, double* out) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double acc = 0.0;
  for (long long k = threadIdx.x; k < len; k += blockDim.x) acc += fabs(X[i * ld + k]);
  acc = wsum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[i] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void add_diag_kernel(int n, double* A, long long ld, double s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(long long)i * ld + i] += s;
}
__global__ void set_identity_kernel(int n, double* A) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long long)n * n) A[t] = ((t / n) == (t % n)) ? 1.0 : 0.0;
}
// out[c*ldo + i] = scale[i] * in[perm[i]*ldi + c]   (gather rows by perm, scale, write TRANSPOSED: vectors -> columns)
__global__ void __launch_bounds__(256) gather_rows_to_cols_kernel(int nrow, long long len, const double* in, long long ldi, const int* perm,
                                                                  const double* scale, double* out, long long ldo) {
  const int i = blockIdx.y;
  const double sc = scale ? scale[i] : 1.0;
  const double* src = in + (long long)perm[i] * ldi;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < len; c += (long long)gridDim.x * blockDim.x)
    out[c * ldo + i] = sc * src[c];
}

static int jacobi_rows(int nvec, int64_t len, double* W, int64_t ldw, double* Vt, double floor2, int* sweeps_out, double stop_below = 1.0e-10) {
  hipStream_t s = hip_stream();
  if (nvec < 2) { if (sweeps_out) *sweeps_out = 0; return QEMB_OK; }
  const int np = (nvec % 2 == 0) ? nvec : nvec + 1;
  // (64 < nvec <= 96 take the block rounds below since round 5: n = 96 cold 3.2-3.4 -> 1.35 ms, warm 1.28 -> 0.60 ms; n = 80 2.1 -> 1.34 / 0.89 -> 0.60 ms -- one
  //  workgroup is bound by the LDS bandwidth of its CU there; at n = 57 the single workgroup is still the faster one, 0.86 vs 0.92 ms.  QEMB_JACOBI_ROWS_SMALL_MAX=96: as before)
  static const int rows_small_max = [] { const char* e = std::getenv("QEMB_JACOBI_ROWS_SMALL_MAX"); return e ? std::atoi(e) : 64; }();
  if (Vt && nvec <= JS_MAX && len <= JS_MAX && nvec <= rows_small_max) {   // LDS-resident single-workgroup path
    int* d_sw = nullptr;
    QTRY_ALLOC(d_sw, sizeof(int));
    const double tol_s = std::max(1.0e-15, std::sqrt((double)len) * 2.22e-16);
    const size_t lds = sizeof(double) * ((size_t)nvec * (len + 1) + (size_t)nvec * (nvec + 1));
    static std::atomic<bool> attr_set{false};   // benign if two threads both set the attribute once
    if (!attr_set) { HIP_TRY(hipFuncSetAttribute((const void*)jacobi_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_set = true; }
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(1), dim3(1024), lds, s, W, (long long)ldw, (int)len, Vt, nvec, np, tol_s, floor2, 40, d_sw);
    HIP_TRY(hipGetLastError());
    int sw = 0;
    HIP_TRY(hipMemcpyAsync(&sw, d_sw, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    (void)dev_free(d_sw);
    if (sweeps_out) *sweeps_out = sw;
    if (sw < 0) { set_error("Jacobi sweeps did not converge in 40 sweeps"); return QEMB_ERR_NOCONV; }
    return QEMB_OK;
  }
  unsigned long long* d_off = nullptr;
  QTRY_ALLOC(d_off, sizeof(unsigned long long));
  const double tol = std::max(1.0e-15, std::sqrt((double)len) * 2.22e-16);   // LAPACK dgesvj-style
  const int max_sweeps = 40;
  // block rounds when two blocks of 16 vectors fit registers + LDS (QEMB_JACOBI_BLOCK=0: per-pair rounds, A/B runs)
  static const bool blocks_enabled = !(std::getenv("QEMB_JACOBI_BLOCK") && std::atoi(std::getenv("QEMB_JACOBI_BLOCK")) == 0);
  int B = 0;
  if (Vt && blocks_enabled && len < (1 << 20)) {
    // two blocks of 16 vectors, [W row | Vt row] <= 600 doubles (10 register slots per lane): eigh for 96 < n <= 300 (the fragment Fock matrices).  (Instantiations with 8 / 4 vectors per block and up to 38 slots were built and measured: an
    // order of magnitude slower than their 16-vector counterpart at the same size and minutes of compile time; longer vectors keep the per-pair rounds.)
    if (nvec >= 32 && len + nvec <= 600) B = 16;
  }
  const int nb = B ? (nvec + B - 1) / B : 0, nbp = (nb % 2 == 0) ? nb : nb + 1;
  const size_t blk_lds = B ? sizeof(double) * 2 * B * (size_t)(len + nvec) : 0;
  const int npl_need = B ? (int)((len + nvec + 63) / 64) : 0;
  // one launch of a block round; the register slots per lane are a template parameter (smallest instantiation that holds a row)
  auto launch_block_round = [&](int r) -> hipError_t {
    hipError_t err = hipSuccess;
    // the three instantiations share one function-pointer type, so a flag inside the generic lambda would be ONE flag for all of them:
    // the dynamic-LDS ceiling is raised on all three, once, and every return code is looked at
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
      const void* ks[3] = {(const void*)jacobi_block_round_kernel<16, 4>, (const void*)jacobi_block_round_kernel<16, 7>, (const void*)jacobi_block_round_kernel<16, 10>};
      for (const void* k : ks) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
      }
    });
    if (attr_err != hipSuccess) return attr_err;
    auto go = [&](auto kern, int bw) {
      if (err == hipSuccess) hipLaunchKernelGGL(kern, dim3(nbp / 2), dim3(bw * 64), blk_lds, s, W, (long long)ldw, (int)len, Vt, nvec, nb, nbp, r, tol, floor2, d_off);
    };
    if (npl_need <= 4) go(jacobi_block_round_kernel<16, 4>, 16); else if (npl_need <= 7) go(jacobi_block_round_kernel<16, 7>, 16); else go(jacobi_block_round_kernel<16, 10>, 16);
    return err;
  };
  int sweep = 0;
  bool conv = false;
  for (; sweep < max_sweeps; ++sweep) {
    HIP_TRY(hipMemsetAsync(d_off, 0, sizeof(unsigned long long), s));
    if (B) {
      for (int r = 0; r < nbp - 1; ++r) HIP_TRY(launch_block_round(r));
    } else
    for (int r = 0; r < np - 1; ++r)
      hipLaunchKernelGGL(jacobi_round_kernel, dim3(np / 2), dim3(256), 0, s, W, (long long)ldw, (long long)len, Vt, nvec, np, r, tol, floor2, d_off);
    HIP_TRY(hipGetLastError());
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, d_off, sizeof(bits), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bits == 0ULL) { conv = true; ++sweep; break; }   // a full sweep without a single rotation
    // Cyclic Jacobi converges quadratically: when the largest normalised off-diagonal element MET during a sweep is eps, what is left
    // after it is O(eps^2).  Below eps = 1e-10 the next sweep would find nothing above `tol` (~3e-15) to rotate -- it would be the
    // confirming no-op sweep, np - 1 launches that change no bit -- so it is not run.
    double offmax; std::memcpy(&offmax, &bits, sizeof(double));
    if (offmax < stop_below) { conv = true; ++sweep; break; }
  }
  (void)dev_free(d_off);
  if (sweeps_out) *sweeps_out = sweep;
  if (!conv) { set_error("Jacobi sweeps did not converge in 40 sweeps"); return QEMB_ERR_NOCONV; }
  return QEMB_OK;
}

int dev_jacobi_eigh(int64_t n64, double* A, double* w, double* V, int* sweeps_out) { return dev_jacobi_eigh_until(n64, A, w, V, sweeps_out, 1.0e-10); }
int dev_jacobi_eigh_until(int64_t n64, double* A, double* w, double* V, int* sweeps_out, double stop_below) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0) return QEMB_OK;
  // small matrices: the whole eigensolve in one launch (two-sided Jacobi in LDS; QEMB_JACOBI_TWOSIDED=0: the one-sided path below, for A/B runs)
  static const bool two_sided = !(std::getenv("QEMB_JACOBI_TWOSIDED") && std::atoi(std::getenv("QEMB_JACOBI_TWOSIDED")) == 0);
  // (n <= 80: the one-barrier kernel; 80 < n <= 96 went through the three-barrier kernel until round 5 -- 1.28 ms warm, bound by the LDS bandwidth of its one CU -- and take
  //  the block rounds of jacobi_rows now, 0.60 ms; QEMB_JACOBI_SMALL_MAX=96: as before)
  static const int small_max = [] { const char* e = std::getenv("QEMB_JACOBI_SMALL_MAX"); return e ? std::min(std::atoi(e), JE_MAX) : JE_DB_MAX; }();
  if (two_sided && n <= small_max) {
    int* d_st = nullptr;
    QTRY_ALLOC(d_st, sizeof(int));
    const int ld = n + 1 + (n & 1), nh = (n + (n & 1)) / 2;
    const size_t lds = sizeof(double) * ((size_t)2 * n * ld + 2 * nh + n) + sizeof(int) * ((size_t)2 * nh + n) + 16;
    static std::atomic<bool> attr_set{false};
    if (!attr_set) { HIP_TRY(hipFuncSetAttribute((const void*)jacobi_eigh_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_set = true; }
    const double tol = std::max(1.0e-15, std::sqrt((double)n) * 2.22e-16);
    static const bool one_barrier = !(std::getenv("QEMB_JACOBI_DB") && std::atoi(std::getenv("QEMB_JACOBI_DB")) == 0);
    if (one_barrier && n <= JE_DB_MAX) {
      const size_t lds_db = sizeof(double) * ((size_t)3 * n * ld + n) + sizeof(int) * (size_t)n + 16;
      static std::atomic<bool> attr_db{false};
      if (!attr_db) { HIP_TRY(hipFuncSetAttribute((const void*)jacobi_eigh_small_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_db = true; }
      const int items = nh * n;                    // <= 4 per thread
      const int nthreads = std::min(1024, std::max(64, (items + 63) / 64 * 64));
      hipLaunchKernelGGL(jacobi_eigh_small_db_kernel, dim3(1), dim3(nthreads), lds_db, s, (const double*)A, n, w, V, tol, stop_below, 40, d_st,
                         (const double*)nullptr, (double*)nullptr, 0, (double*)nullptr);
    } else {
    const int nthreads = n <= 32 ? 256 : (n <= 64 ? 512 : 1024);
    hipLaunchKernelGGL(jacobi_eigh_small_kernel, dim3(1), dim3(nthreads), lds, s, (const double*)A, n, w, V, tol, stop_below, 40, d_st);
    }
    HIP_TRY(hipGetLastError());
    int st = 0;
    HIP_TRY(hipMemcpyAsync(&st, d_st, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    (void)dev_free(d_st);
    if (sweeps_out) *sweeps_out = st;
    if (st < 0) { set_error("Jacobi sweeps did not converge in 40 sweeps"); return QEMB_ERR_NOCONV; }
    return QEMB_OK;
  }
  double *Vt = nullptr, *tmp = nullptr; int* d_perm = nullptr;
  QTRY_ALLOC(Vt, sizeof(double) * (size_t)n * n);
  QTRY_ALLOC(tmp, sizeof(double) * (size_t)n);
  QTRY_ALLOC(d_perm, sizeof(int) * (size_t)n);
  // Gershgorin shift
  hipLaunchKernelGGL(rowabssum_kernel, dim3(n), dim3(256), 0, s, n, (long long)n, A, (long long)n, tmp);
  std::vector<double> h(n);
  HIP_TRY(hipMemcpyAsync(h.data(), tmp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  double gersh = 0.0;
  for (double x : h) gersh = std::max(gersh, x);
  const double sigma = 1.0625 * gersh + 1.0e-300;
  hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, A, (long long)n, sigma);
  hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)(((long long)n * n + 255) / 256)), dim3(256), 0, s, n, Vt);
  int rc = jacobi_rows(n, n, A, n, Vt, 0.0, sweeps_out, stop_below);
  if (rc == QEMB_OK) {
    // Rayleigh quotients lambda_i + sigma = W_i . Vt_i
    hipLaunchKernelGGL(rowdot_kernel, dim3(n), dim3(256), 0, s, n, (long long)n, A, (long long)n, Vt, (long long)n, tmp);
    HIP_TRY(hipMemcpyAsync(h.data(), tmp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return h[a] < h[b]; });
    std::vector<double> ws(n);
    for (int i = 0; i < n; ++i) ws[i] = h[perm[i]] - sigma;
    HIP_TRY(hipMemcpyAsync(w, ws.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gather_rows_to_cols_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, n, (long long)n, Vt, (long long)n, d_perm, (const double*)nullptr, V, (long long)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
  }
  (void)dev_free(Vt); (void)dev_free(tmp); (void)dev_free(d_perm);
  return rc;
}

// ---- fused steps of the fragment RHF of small fragments (scf.cpp)
int dev_scf_fused_max() {
  static const int m = [] { const char* e = std::getenv("QEMB_SCF_FUSED"); return (e && std::atoi(e) == 0) ? 0 : JE_DB_MAX; }();      // (QEMB_SCF_FUSED=0: the launch-by-launch cycle, for A/B runs)
  return m;
}
int dev_jacobi_eigh_in_basis(int64_t n64, const double* F, const double* Cp, double* w, double* C_out, double* C2_out, int nocc, double* dm_out, double stop_below,
                             int* status_dev) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0 || n > JE_DB_MAX || nocc < 0 || nocc > n || !status_dev) { set_error("dev_jacobi_eigh_in_basis: 0 < n <= 80, 0 <= nocc <= n, a status word"); return QEMB_ERR_ARG; }
  const int ld = n + 1 + (n & 1), nh = (n + (n & 1)) / 2;
  const size_t lds_db = sizeof(double) * ((size_t)3 * n * ld + n) + sizeof(int) * (size_t)n + 16;
  static std::atomic<bool> attr_db{false};
  if (!attr_db) { HIP_TRY(hipFuncSetAttribute((const void*)jacobi_eigh_small_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_db = true; }
  const double tol = std::max(1.0e-15, std::sqrt((double)n) * 2.22e-16);
  const int items = nh * n;
  const int nthreads = std::min(1024, std::max(64, (items + 63) / 64 * 64));
  hipLaunchKernelGGL(jacobi_eigh_small_db_kernel, dim3(1), dim3(nthreads), lds_db, s, F, n, w, C_out, tol, stop_below, 40, status_dev, Cp, C2_out, nocc, dm_out);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_scf_fock_small(int64_t n64, const double* h, const double* J, const double* K, const double* D, double* F, double* err, double* scal2) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0 || n > JE_DB_MAX) { set_error("dev_scf_fock_small: 0 < n <= 80"); return QEMB_ERR_ARG; }
  const int ld = n + 1 + (n & 1);
  const int nthreads = n <= 24 ? 256 : (n <= 40 ? 512 : 1024);
  const size_t lds = sizeof(double) * ((size_t)2 * n * ld + 2 * nthreads);
  static std::atomic<bool> attr{false};
  if (!attr) { HIP_TRY(hipFuncSetAttribute((const void*)scf_fock_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr = true; }
  hipLaunchKernelGGL(scf_fock_small_kernel, dim3(1), dim3(nthreads), lds, s, n, h, J, K, D, F, err, scal2);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}
int dev_pack_density_sym(int64_t n64, const double* D, double* Dp) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0) return QEMB_OK;
  const long long np = (long long)n * (n + 1) / 2;
  hipLaunchKernelGGL(pack_density_sym_kernel, dim3((unsigned)std::min<long long>((np + 255) / 256, 4096)), dim3(256), 0, s, n, D, Dp);
  HIP_TRY(hipGetLastError());
  return QEMB_OK;
}

int dev_jacobi_svd(int64_t m64, int64_t n64, double* G, double* sv, double* U, double* V, int* sweeps_out) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64; const long long m = m64;
  if (n <= 0 || m <= 0) return QEMB_OK;
  if (m < n) { set_error("dev_jacobi_svd: need m >= n"); return QEMB_ERR_ARG; }
  double *Wt = nullptr, *Vt = nullptr, *tmp = nullptr; int* d_perm = nullptr;
  QTRY_ALLOC(Wt, sizeof(double) * (size_t)n * m);
  QTRY_ALLOC(Vt, sizeof(double) * (size_t)n * n);
  QTRY_ALLOC(tmp, sizeof(double) * (size_t)n);
  QTRY_ALLOC(d_perm, sizeof(int) * (size_t)n);
  // Wt = G^T (n x m): columns of G become contiguous rows
  Copy4Desc c{};
  c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = n; c.dim[3] = m;
  c.in = G; c.si[0] = 0; c.si[1] = 0; c.si[2] = 1; c.si[3] = n;
  c.out = Wt; c.so[0] = 0; c.so[1] = 0; c.so[2] = m; c.so[3] = 1;
  c.alpha = 1.0; c.beta = 0.0;
  int rc = dev_copy4(c);
  if (rc == QEMB_OK) {
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)(((long long)n * n + 255) / 256)), dim3(256), 0, s, n, Vt);
    // scale for the "zero vector" floor: ||G||_F^2
    hipLaunchKernelGGL(rowdot_kernel, dim3(n), dim3(256), 0, s, n, m, Wt, m, Wt, m, tmp);
    std::vector<double> h(n);
    HIP_TRY(hipMemcpyAsync(h.data(), tmp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    double fro2 = 0.0;
    for (double x : h) fro2 += x;
    const double floor2 = fro2 * 1.0e-30;   // vectors below 1e-15 * ||G||_F are numerically zero
    rc = jacobi_rows(n, m, Wt, m, Vt, floor2, sweeps_out);
    if (rc == QEMB_OK) {
      hipLaunchKernelGGL(rowdot_kernel, dim3(n), dim3(256), 0, s, n, m, Wt, m, Wt, m, tmp);
      HIP_TRY(hipMemcpyAsync(h.data(), tmp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      std::vector<int> perm(n);
      std::iota(perm.begin(), perm.end(), 0);
      std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return h[a] > h[b]; });
      std::vector<double> ss(n), inv(n);
      for (int i = 0; i < n; ++i) {
        ss[i] = std::sqrt(std::max(h[perm[i]], 0.0));
        inv[i] = (h[perm[i]] > floor2) ? 1.0 / ss[i] : 0.0;
      }
      HIP_TRY(hipMemcpyAsync(sv, ss.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(tmp, inv.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * n, hipMemcpyHostToDevice, s));
      if (U) hipLaunchKernelGGL(gather_rows_to_cols_kernel, dim3((unsigned)std::min<long long>((m + 255) / 256, 65535), n), dim3(256), 0, s, n, m, Wt, m, d_perm, tmp, U, (long long)n);
      if (V) hipLaunchKernelGGL(gather_rows_to_cols_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, n, (long long)n, Vt, (long long)n, d_perm, (const double*)nullptr, V, (long long)n);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(s));
    }
  }
  (void)dev_free(Wt); (void)dev_free(Vt); (void)dev_free(tmp); (void)dev_free(d_perm);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Cholesky / triangular inverse
// ------------------------------------------------------------------------------------------------
constexpr int NB = 32;

// Factor the nb x nb diagonal block at A (ld) in place (lower), write its inverse to Dinv (NB x NB, ld NB).
// Also used with factor=false to only invert an existing lower-triangular block.
__global__ void __launch_bounds__(256) diag_block_kernel(double* A, long long ld, int nb, double* Dinv, int factor, int* fail) {
  __shared__ double a[NB][NB + 1];
  __shared__ double inv[NB][NB + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int t = tid; t < NB * NB; t += 256) {
    const int i = t / NB, j = t % NB;
    a[i][j] = (i < nb && j < nb && j <= i) ? A[(long long)i * ld + j] : ((i == j) ? 1.0 : 0.0);
    inv[i][j] = 0.0;
  }
  __syncthreads();
  if (factor) {
    for (int k = 0; k < nb; ++k) {
      if (tid == 0) {
        const double d = a[k][k];
        if (!(d > 0.0)) { bad = 1; a[k][k] = 1.0; } else a[k][k] = sqrt(d);
      }
      __syncthreads();
      const double dk = a[k][k];
      for (int i = k + 1 + tid; i < nb; i += 256) a[i][k] /= dk;
      __syncthreads();
      for (int t = tid; t < NB * NB; t += 256) {
        const int i = t / NB, j = t % NB;
        if (i > k && j > k && j <= i && i < nb) a[i][j] -= a[i][k] * a[j][k];
      }
      __syncthreads();
    }
    for (int t = tid; t < NB * NB; t += 256) {
      const int i = t / NB, j = t % NB;
      if (i < nb && j < nb) A[(long long)i * ld + j] = (j <= i) ? a[i][j] : 0.0;
    }
  }
  __syncthreads();
  if (tid < nb) {   // column tid of the inverse by forward substitution
    const int j = tid;
    inv[j][j] = 1.0 / a[j][j];
    for (int i = j + 1; i < nb; ++i) {
      double acc = 0.0;
      for (int k = j; k < i; ++k) acc += a[i][k] * inv[k][j];
      inv[i][j] = -acc / a[i][i];
    }
  }
  __syncthreads();
  for (int t = tid; t < NB * NB; t += 256) Dinv[t] = inv[t / NB][t % NB];
  if (tid == 0 && bad && fail) *fail = 1;
}
__global__ void zero_upper_kernel(int n, double* A) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long long)n * n) { const long long i = t / n, j = t % n; if (j > i) A[t] = 0.0; }
}

int dev_cholesky_lower(int64_t n64, double* A) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0) return QEMB_OK;
  double *Dinv = nullptr, *panel = nullptr; int* d_fail = nullptr;
  QTRY_ALLOC(Dinv, sizeof(double) * NB * NB);
  QTRY_ALLOC(panel, sizeof(double) * (size_t)n * NB);
  QTRY_ALLOC(d_fail, sizeof(int));
  HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(int), s));
  int rc = QEMB_OK;
  for (int k0 = 0; k0 < n && rc == QEMB_OK; k0 += NB) {
    const int nb = std::min(NB, n - k0);
    double* Akk = A + (long long)k0 * n + k0;
    hipLaunchKernelGGL(diag_block_kernel, dim3(1), dim3(256), 0, s, Akk, (long long)n, nb, Dinv, 1, d_fail);
    const int rest = n - k0 - nb;
    if (rest > 0) {
      double* A21 = A + (long long)(k0 + nb) * n + k0;
      // panel = A21 * L11^{-T}:  B(k,j) = Dinv[j][k]  -> stored N x K  (b_kcontig)
      GemmDesc g{};
      g.M = rest; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
      g.A = A21; g.lda = n; g.a_kcontig = 1; g.strideA = 0;
      g.B = Dinv; g.ldb = NB; g.b_kcontig = 1; g.strideB = 0;
      g.C = panel; g.ldc = NB; g.strideC = 0; g.batch = 1;
      rc = dev_gemm(g);
      if (rc) break;
      Copy4Desc c{};
      c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = rest; c.dim[3] = nb;
      c.in = panel; c.si[0] = 0; c.si[1] = 0; c.si[2] = NB; c.si[3] = 1;
      c.out = A21; c.so[0] = 0; c.so[1] = 0; c.so[2] = n; c.so[3] = 1; c.alpha = 1.0; c.beta = 0.0;
      rc = dev_copy4(c);
      if (rc) break;
      // A22 -= L21 L21^T
      double* A22 = A + (long long)(k0 + nb) * n + (k0 + nb);
      GemmDesc u{};
      u.M = rest; u.N = rest; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
      u.A = panel; u.lda = NB; u.a_kcontig = 1; u.strideA = 0;
      u.B = panel; u.ldb = NB; u.b_kcontig = 1; u.strideB = 0;
      u.C = A22; u.ldc = n; u.strideC = 0; u.batch = 1;
      rc = dev_gemm(u);
    }
  }
  if (rc == QEMB_OK) {
    hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)(((long long)n * n + 255) / 256)), dim3(256), 0, s, n, A);
    int fail = 0;
    HIP_TRY(hipMemcpyAsync(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (fail) { set_error("Cholesky: matrix is not positive definite"); rc = QEMB_ERR_NUMERIC; }
  }
  (void)dev_free(Dinv); (void)dev_free(panel); (void)dev_free(d_fail);
  return rc;
}

int dev_tri_inverse_lower(int64_t n64, const double* L, double* X) {
  hipStream_t s = hip_stream();
  if (!s) { set_error("libqemb_hip: call qemb_init(device) first"); return QEMB_ERR_DEVICE; }
  const int n = (int)n64;
  if (n <= 0) return QEMB_OK;
  double *Dinv = nullptr, *T = nullptr, *Lc = nullptr;
  QTRY_ALLOC(Dinv, sizeof(double) * NB * NB);
  QTRY_ALLOC(T, sizeof(double) * (size_t)NB * n);
  QTRY_ALLOC(Lc, sizeof(double) * (size_t)n * n);   // diag_block_kernel takes non-const
  HIP_TRY(hipMemcpyAsync(Lc, L, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, s));
  int rc = dev_fill(X, (int64_t)n * n, 0.0);
  for (int i0 = 0; i0 < n && rc == QEMB_OK; i0 += NB) {
    const int nb = std::min(NB, n - i0);
    hipLaunchKernelGGL(diag_block_kernel, dim3(1), dim3(256), 0, s, Lc + (long long)i0 * n + i0, (long long)n, nb, Dinv, 0, (int*)nullptr);
    // X[i,i] = Dinv
    Copy4Desc c{};
    c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = nb; c.dim[3] = nb;
    c.in = Dinv; c.si[0] = 0; c.si[1] = 0; c.si[2] = NB; c.si[3] = 1;
    c.out = X + (long long)i0 * n + i0; c.so[0] = 0; c.so[1] = 0; c.so[2] = n; c.so[3] = 1; c.alpha = 1.0; c.beta = 0.0;
    rc = dev_copy4(c);
    if (rc || i0 == 0) continue;
    // T (nb x i0) = L[i, 0:i0] * X[0:i0, 0:i0]
    GemmDesc g{};
    g.M = nb; g.N = i0; g.K = i0; g.alpha = 1.0; g.beta = 0.0;
    g.A = Lc + (long long)i0 * n; g.lda = n; g.a_kcontig = 1; g.strideA = 0;
    g.B = X; g.ldb = n; g.b_kcontig = 0; g.strideB = 0;
    g.C = T; g.ldc = n; g.strideC = 0; g.batch = 1;
    rc = dev_gemm(g);
    if (rc) break;
    // X[i, 0:i0] = -Dinv * T
    GemmDesc h{};
    h.M = nb; h.N = i0; h.K = nb; h.alpha = -1.0; h.beta = 0.0;
    h.A = Dinv; h.lda = NB; h.a_kcontig = 1; h.strideA = 0;
    h.B = T; h.ldb = n; h.b_kcontig = 0; h.strideB = 0;
    h.C = X + (long long)i0 * n; h.ldc = n; h.strideC = 0; h.batch = 1;
    rc = dev_gemm(h);
  }
  HIP_TRY(hipStreamSynchronize(s));
  (void)dev_free(Dinv); (void)dev_free(T); (void)dev_free(Lc);
  return rc;
}

}  // namespace qemb
