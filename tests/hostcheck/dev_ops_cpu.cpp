#include <atomic>
// dev_ops_cpu.cpp -- TEST-ONLY scalar mock of quemb_amd/csrc/dev_ops.h.
//
// Purpose: let the GPU-less build container exercise the HOST LOGIC of the drivers (ccsd.cpp, scf.cpp,
// ao2mo.cpp, schmidt.cpp, fragment.cpp: the GEMM factorisation of the CCSD equations, index permutations,
// DIIS, convergence control) against the oracle.  "Device" memory is plain host memory here.
// This file is linked ONLY into tests/hostcheck/libqemb_hostcheck.so by tests/hostcheck/build.py; it is never
// part of libqemb_hip.so, never imported by quemb_amd, and proves nothing about the HIP kernels (those are
// checked on the GPU by tests/test_gpu_*.py).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include "dev_ops.h"

namespace qemb {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
const char* dev_backend_name() { return "hostcheck"; }
int dev_gemm_stamps(const GemmDesc&, int, double*) { set_error("dev_gemm_stamps: not available in the hostcheck build"); return QEMB_ERR_DEVICE; }
void dev_gemm_set_force_cfg(int) {}
void dev_gemm_set_auto_splitk(int) {}
void dev_gemm_set_peers(int) {}
int dev_gemm_peers() { return 1; }

int dev_init(int) { return 0; }
int dev_sync() { return 0; }
int dev_sync_device() { return QEMB_OK; }
int dev_alloc(void** p, size_t bytes) { *p = std::malloc(bytes ? bytes : 16); if (!*p) { set_error("malloc failed"); return QEMB_ERR_ALLOC; } return 0; }
int dev_free(void* p) { std::free(p); return 0; }
int dev_trim() { return 0; }
int dev_trim_all() { return dev_trim(); }
int dev_h2d(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return 0; }
int dev_h2d_async(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return 0; }
int dev_d2h(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return 0; }
int dev_d2h_async(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return 0; }
int dev_pinned_alloc(void** p, size_t b) { *p = std::malloc(b ? b : 16); return *p ? 0 : QEMB_ERR_ALLOC; }
int dev_pinned_free(void* p) { std::free(p); return 0; }
int dev_d2d(void* d, const void* s, size_t b) { std::memmove(d, s, b); return 0; }
int dev_fill(double* x, int64_t n, double v) { std::fill(x, x + n, v); return 0; }
int dev_graph_begin(int) { return 1; }
int dev_region_begin() { return 0; }
int dev_region_chain() { return 0; }
int dev_region_end() { return 0; }   // the mock cannot capture: drivers run eagerly
int dev_graph_end(dev_graph_t*) { return QEMB_ERR_DEVICE; }
int dev_graph_launch(dev_graph_t) { return QEMB_ERR_DEVICE; }
int dev_graph_destroy(dev_graph_t) { return 0; }
int dev_tape_end(dev_tape_t*) { return 1; }            // the mock executes eagerly: nothing to tape
int dev_tape_equal(dev_tape_t, dev_tape_t) { return 0; }
void dev_alloc_trace_begin() {}
unsigned long long dev_alloc_trace_end() { return 0; }
int dev_tape_run(const dev_tape_t*, int) { set_error("hostcheck: no tapes"); return QEMB_ERR_DEVICE; }
int dev_tape_destroy(dev_tape_t) { return QEMB_OK; }
int dev_tape_last_stats(long long* a, long long* b, long long* c) { if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; return QEMB_OK; }
bool dev_capturing() { return false; }
int dev_mem_info(size_t* f, size_t* t) { *f = *t = (size_t)1 << 34; return 0; }
int dev_alloc_stats(long long* n, long long* nfree, double* ms, double* gb, int) { if (n) *n = 0; if (nfree) *nfree = 0; if (ms) *ms = 0.0; if (gb) *gb = 0.0; return 0; }

static double g_tot[TIMER_NSLOTS]; static int64_t g_cnt[TIMER_NSLOTS];
static std::chrono::steady_clock::time_point g_t0[TIMER_NSLOTS];
int dev_timer_begin(int s) { g_t0[s] = std::chrono::steady_clock::now(); return 0; }
int dev_timer_end(int s) { g_tot[s] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0[s]).count(); g_cnt[s]++; return 0; }
int dev_timer_read(int s, double* ms, int64_t* c) { if (ms) *ms = g_tot[s]; if (c) *c = g_cnt[s]; return 0; }
int dev_timer_reset(int s) { g_tot[s] = 0; g_cnt[s] = 0; return 0; }
int dev_timer_live_events(int) { return 0; }

int dev_gemm_probe(const GemmDesc&, double*, double*, long long*) { set_error("dev_gemm_probe: not available in the hostcheck build"); return QEMB_ERR_DEVICE; }
static std::atomic<long long> g_gemm_flops{0};
int dev_gemm_flop_count(double* flops, int reset) { if (flops) *flops = (double)g_gemm_flops.load(); if (reset) g_gemm_flops = 0; return QEMB_OK; }
int dev_gemm(const GemmDesc& g0) {
  g_gemm_flops += 2ll * g0.M * g0.N * g0.K * g0.batch;
  GemmDesc g = g0;
  if (g.keep_slabs) {      // slab form: the first slab receives the whole product, the others zeros (their sum is what the consumer forms)
    if (g.ksplit <= 1 || g.alpha != 1.0 || g.beta != 0.0 || g.batch != 1) { set_error("dev_gemm: keep_slabs needs ksplit > 1, alpha = 1, beta = 0, batch = 1"); return QEMB_ERR_ARG; }
    const int S = gemm_slab_count(g.K, g.ksplit);
    std::fill(g.C + g.M * g.N, g.C + (int64_t)S * g.M * g.N, 0.0);
    g.ldc = g.N;
  }
  for (int64_t b = 0; b < g.batch; ++b) {
    const double* A = g.A + b * g.strideA; const double* B = g.B + b * g.strideB; double* C = g.C + b * g.strideC;
    // pack to contiguous row-major A(MxK), B(KxN) for a cache-friendly triple loop
    std::vector<double> a((size_t)g.M * g.K), bb((size_t)g.K * g.N);
    if (g.a_slab > 0 && (g.a_kcontig || g.batch != 1)) { set_error("dev_gemm: a_slab needs a !a_kcontig A operand and batch = 1"); return QEMB_ERR_ARG; }
    for (int64_t m = 0; m < g.M; ++m) for (int64_t k = 0; k < g.K; ++k)
      a[m * g.K + k] = g.a_kcontig ? A[m * g.lda + k] : A[k * g.lda + m + (g.a_slab > 0 ? (m / g.a_slab) * g.a_slab_skip : 0)];
    for (int64_t k = 0; k < g.K; ++k) for (int64_t n = 0; n < g.N; ++n) bb[k * g.N + n] = g.b_kcontig ? B[n * g.ldb + k] : B[k * g.ldb + n];
    std::vector<double> row((size_t)g.N);
    for (int64_t m = 0; m < g.M; ++m) {
      std::fill(row.begin(), row.end(), 0.0);
      for (int64_t k = 0; k < g.K; ++k) { const double x = a[m * g.K + k]; const double* br = &bb[k * g.N]; for (int64_t n = 0; n < g.N; ++n) row[n] += x * br[n]; }
      for (int64_t n = 0; n < g.N; ++n) { double* c = C + m * g.ldc + n; *c = (g.beta != 0.0) ? g.alpha * row[n] + g.beta * (*c) : g.alpha * row[n]; }
    }
  }
  return 0;
}

int dev_copy4(const Copy4Desc& c) {
  for (int64_t i0 = 0; i0 < c.dim[0]; ++i0) for (int64_t i1 = 0; i1 < c.dim[1]; ++i1) for (int64_t i2 = 0; i2 < c.dim[2]; ++i2) for (int64_t i3 = 0; i3 < c.dim[3]; ++i3) {
    const double v = c.alpha * c.in[i0 * c.si[0] + i1 * c.si[1] + i2 * c.si[2] + i3 * c.si[3]];
    const int64_t off = i0 * c.so[0] + i1 * c.so[1] + i2 * c.so[2] + i3 * c.so[3];
    const double* base = c.base ? c.base : c.out;
    const double w = (c.beta != 0.0) ? v + c.beta * base[off] : v;
    c.out[off] = w;
    if (c.out2) c.out2[off] = c.c2a * c.in2[off] + c.c2b * w;
  }
  return 0;
}
int dev_outer4(const Outer4Desc& c) {
  for (int64_t i0 = 0; i0 < c.dim[0]; ++i0) for (int64_t i1 = 0; i1 < c.dim[1]; ++i1) for (int64_t i2 = 0; i2 < c.dim[2]; ++i2) for (int64_t i3 = 0; i3 < c.dim[3]; ++i3) {
    const double v = c.alpha * c.u[i0 * c.su0 + i2 * c.su2] * c.v[i1 * c.sv1 + i3 * c.sv3];
    const int64_t off = i0 * c.so[0] + i1 * c.so[1] + i2 * c.so[2] + i3 * c.so[3];
    const double* base = c.base ? c.base : c.out;
    c.out[off] = (c.beta != 0.0) ? v + c.beta * base[off] : v;
  }
  return 0;
}
int dev_div_denom(double* x, int64_t d0, int64_t d1, int64_t d2, int64_t d3, const double* ea, const double* eb, const double* ec, const double* ed) {
  for (int64_t i0 = 0; i0 < d0; ++i0) for (int64_t i1 = 0; i1 < d1; ++i1) for (int64_t i2 = 0; i2 < d2; ++i2) for (int64_t i3 = 0; i3 < d3; ++i3)
    x[((i0 * d1 + i1) * d2 + i2) * d3 + i3] /= (ea[i0] + (eb ? eb[i1] : 0.0) - ec[i2] - (ed ? ed[i3] : 0.0));
  return 0;
}
int dev_ladder_pack_vvvv(int64_t n, int64_t o, const double* M, double* Vp, int64_t ldp, double* Vm, int64_t ldm) {
  const int64_t v = n - o;
  for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b <= a; ++b) {
    double* vp = Vp + (a * (a + 1) / 2 + b) * ldp; std::fill(vp, vp + ldp, 0.0);
    double* vm = a > b ? Vm + (a * (a - 1) / 2 + b) * ldm : nullptr; if (vm) std::fill(vm, vm + ldm, 0.0);
    for (int64_t c = 0; c < v; ++c) for (int64_t d = 0; d <= c; ++d) {
      const double x = M[(((o + a) * n + (o + c)) * n + (o + b)) * n + (o + d)], y = M[(((o + a) * n + (o + d)) * n + (o + b)) * n + (o + c)];
      vp[c * (c + 1) / 2 + d] = x + y;
      if (vm && c > d) vm[c * (c - 1) / 2 + d] = x - y;
    }
  }
  return 0;
}
static inline int64_t pidx(int64_t i, int64_t j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
int dev_lincomb(int64_t n, int nterms, const double* coef, const double* const* xs, double beta, double* out) {
  for (int64_t t = 0; t < n; ++t) { double acc = beta != 0.0 ? beta * out[t] : 0.0; for (int q = 0; q < nterms; ++q) acc += coef[q] * xs[q][t]; out[t] = acc; }
  return 0;
}
int dev_ctx_timer_read(int k, int slot, double* total_ms, int64_t* count, int reset) { (void)k; (void)reset; return dev_timer_read(slot, total_ms, count); }
int dev_ctx_count(int n) { (void)n; return 1; }
int dev_ctx_bind(int k) { return k == 0 ? 0 : QEMB_ERR_ARG; }
int dev_ctx_partition(int parts) { return (parts < 0 || parts > 8) ? QEMB_ERR_ARG : 0; }
int dev_mirror_lower(int64_t n, double* A, int64_t lda) {
  for (int64_t r = 0; r < n; ++r) for (int64_t c = r + 1; c < n; ++c) A[r * lda + c] = A[c * lda + r];
  return 0;
}
int dev_k_from_pairs(int64_t n, const double* H, const double* D, double* K) {
  std::fill(K, K + n * n, 0.0);
  for (int64_t p = 0; p < n; ++p) for (int64_t q = 0; q <= p; ++q) {
    const double* X = H + pidx(p, q) * n * n;
    for (int64_t r = 0; r < n; ++r) {
      double a1 = 0.0, a2 = 0.0;
      for (int64_t s = 0; s < n; ++s) { a1 += X[r * n + s] * D[q * n + s]; a2 += X[r * n + s] * D[p * n + s]; }
      K[p * n + r] += a1;
      if (p != q) K[q * n + r] += a2;
    }
  }
  return 0;
}
int dev_jk_from_packed(int64_t n, const double* S4, const double* D, const double* Dp, double* Jp, double* K) {
  if (n <= 0) return 0;
  if (n > 1024) { set_error("dev_jk_from_packed: n > 1024 (use dev_k_from_pairs)"); return QEMB_ERR_ARG; }
  if (!S4 || !D || (Jp && !Dp) || (!Jp && !K)) { set_error("dev_jk_from_packed: bad arguments"); return QEMB_ERR_ARG; }
  const int64_t np = n * (n + 1) / 2;
  if (K) std::fill(K, K + n * n, 0.0);
  for (int64_t p = 0; p < n; ++p) for (int64_t q = 0; q <= p; ++q) {
    const double* row = S4 + pidx(p, q) * np;
    if (Jp) { double a = 0.0; for (int64_t i = 0; i < np; ++i) a += row[i] * Dp[i]; Jp[pidx(p, q)] = a; }
    if (!K) continue;
    for (int64_t r = 0; r < n; ++r) {
      double a1 = 0.0, a2 = 0.0;
      for (int64_t s = 0; s < n; ++s) { const double x = row[pidx(r, s)]; a1 += x * D[q * n + s]; a2 += x * D[p * n + s]; }
      K[p * n + r] += a1;
      if (p != q) K[q * n + r] += a2;
    }
  }
  return 0;
}
int dev_pack_pair_rows(int64_t n, int64_t ncols, const double* in, double* out) {
  for (int64_t x = 0; x < n; ++x) for (int64_t y = 0; y <= x; ++y) std::copy(in + (x * n + y) * ncols, in + (x * n + y + 1) * ncols, out + pidx(x, y) * ncols);
  return 0;
}
int dev_extract_pf(int64_t n, const double* Mp, int64_t p0, int64_t q0, int64_t r0, int64_t s0, int64_t sp, int64_t sq, int64_t sr, int64_t ss, double* out) {
  for (int64_t p = 0; p < sp; ++p) for (int64_t q = 0; q < sq; ++q) for (int64_t r = 0; r < sr; ++r) for (int64_t s = 0; s < ss; ++s)
    out[((p * sq + q) * sr + r) * ss + s] = Mp[(pidx(p0 + p, q0 + q) * n + (r0 + r)) * n + (s0 + s)];
  return 0;
}
int dev_extract_pf_t(int64_t n, const double* T, int64_t x0, int64_t r0, int64_t s0, int64_t c0, int64_t sx, int64_t sr, int64_t ss, int64_t sc, double* out, int64_t slab) {
  if (slab <= 0) slab = n * n;
  for (int64_t x = 0; x < sx; ++x) for (int64_t r = 0; r < sr; ++r) for (int64_t s = 0; s < ss; ++s) for (int64_t c = 0; c < sc; ++c)
    out[((x * sr + r) * ss + s) * sc + c] = T[pidx(r0 + r, s0 + s) * slab + (c0 + c) * n + (x0 + x)];
  return 0;
}
int dev_ladder_pack_vvvv_pf(int64_t n, int64_t o, const double* Mp, double* Vp, int64_t ldp, double* Vm, int64_t ldm) {
  const int64_t v = n - o;
  for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b <= a; ++b) {
    double* vp = Vp + (a * (a + 1) / 2 + b) * ldp; std::fill(vp, vp + ldp, 0.0);
    double* vm = a > b ? Vm + (a * (a - 1) / 2 + b) * ldm : nullptr; if (vm) std::fill(vm, vm + ldm, 0.0);
    for (int64_t c = 0; c < v; ++c) for (int64_t d = 0; d <= c; ++d) {
      const double x = Mp[(pidx(o + a, o + c) * n + (o + b)) * n + (o + d)], y = Mp[(pidx(o + b, o + c) * n + (o + a)) * n + (o + d)];
      vp[c * (c + 1) / 2 + d] = x + y;
      if (vm && c > d) vm[c * (c - 1) / 2 + d] = x - y;
    }
  }
  return 0;
}
int dev_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm) {
  for (int64_t r = 0; r < rows; ++r) {
    const double* t = in + r * v * v;
    double* tp = Op + r * ldp; std::fill(tp, tp + ldp, 0.0);
    double* tm = Om + r * ldm; std::fill(tm, tm + ldm, 0.0);
    for (int64_t c = 0; c < v; ++c) for (int64_t d = 0; d <= c; ++d) {
      tp[c * (c + 1) / 2 + d] = t[c * v + d] + t[d * v + c];
      if (c > d) tm[c * (c - 1) / 2 + d] = t[c * v + d] - t[d * v + c];
    }
  }
  return 0;
}
int dev_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out, const double* add, int Sp, int64_t strideP, int Sm, int64_t strideM) {
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j <= i; ++j) for (int64_t c = 0; c < ncols; ++c) {
    double p = 0.0;
    for (int sl = 0; sl < std::max(Sp, 1); ++sl) p += Xp[sl * strideP + (i * (i + 1) / 2 + j) * ncols + c];
    const int64_t ij = (i * o + j) * ncols + c, ji = (j * o + i) * ncols + c;
    if (i > j) { double m = 0.0; for (int sl = 0; sl < std::max(Sm, 1); ++sl) m += Xm[sl * strideM + (i * (i - 1) / 2 + j) * ncols + c]; out[ij] = (add ? add[ij] : 0.0) + (p + m); out[ji] = (add ? add[ji] : 0.0) + (p - m); }
    else out[ij] = (add ? add[ij] : 0.0) + p;
  }
  return 0;
}
int dev_ladder_pack_tau(int64_t o, int64_t v, const double* tau, double* Tp, int64_t ldp, double* Tm, int64_t ldm) {
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j <= i; ++j) {
    const double* t = tau + (i * o + j) * v * v;
    double* tp = Tp + (i * (i + 1) / 2 + j) * ldp; std::fill(tp, tp + ldp, 0.0);
    double* tm = i > j ? Tm + (i * (i - 1) / 2 + j) * ldm : nullptr; if (tm) std::fill(tm, tm + ldm, 0.0);
    for (int64_t c = 0; c < v; ++c) for (int64_t d = 0; d <= c; ++d) {
      const double x = t[c * v + d], y = t[d * v + c];
      tp[c * (c + 1) / 2 + d] = (c == d ? 0.25 : 0.5) * (x + y);
      if (tm && c > d) tm[c * (c - 1) / 2 + d] = 0.5 * (x - y);
    }
  }
  return 0;
}
int dev_ladder_scatter_pm2(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, const double* Hp, const double* Hm,
                           int assign, double* t2, int Sp, int64_t strideP, int Sm, int64_t strideM, int64_t ldhp, int64_t ldhm) {
  if (assign) std::fill(t2, t2 + o * o * v * v, 0.0);
  if (!ldhp) ldhp = ldp;
  if (!ldhm) ldhm = ldm;
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j <= i; ++j) for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b <= a; ++b) {
    const bool has_m = (i > j && a > b);
    const int64_t cp = a * (a + 1) / 2 + b, cm = a * (a - 1) / 2 + b, rp = i * (i + 1) / 2 + j, rm = i * (i - 1) / 2 + j;
    double p = 0.0, m = 0.0;
    for (int sl = 0; sl < std::max(Sp, 1); ++sl) p += Rp[sl * strideP + rp * ldp + cp];
    if (has_m) for (int sl = 0; sl < std::max(Sm, 1); ++sl) m += Rm[sl * strideM + rm * ldm + cm];
    if (Hp) p += (a == b ? 2.0 : 1.0) * Hp[rp * ldhp + cp];
    if (Hm && has_m) m += Hm[rm * ldhm + cm];
    t2[((i * o + j) * v + a) * v + b] += p + m;
    if (a != b) t2[((i * o + j) * v + b) * v + a] += p - m;
    if (i != j) { t2[((j * o + i) * v + a) * v + b] += p - m; if (a != b) t2[((j * o + i) * v + b) * v + a] += p + m; }
  }
  return 0;
}
int dev_ladder_scatter_pm(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, double* t2) {
  return dev_ladder_scatter_pm2(o, v, Rp, ldp, Rm, ldm, nullptr, nullptr, 0, t2);
}
int dev_foo_from_x(int64_t o, const double* X, double* F) {
  for (int64_t k = 0; k < o; ++k) for (int64_t i = 0; i < o; ++i) {
    double s = 0.0;
    for (int64_t l = 0; l < o; ++l) s += 2.0 * X[((i * o + l) * o + k) * o + l] - X[((l * o + i) * o + k) * o + l];
    F[k * o + i] = s;
  }
  return 0;
}
int dev_pack_w_pm(int64_t o, const double* W, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) {
  const int64_t npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2;
  auto w = [&](int64_t k, int64_t l, int64_t i, int64_t j) { return W[((k * o + l) * o + i) * o + j]; };
  std::fill(Ap, Ap + npo * lda_p, 0.0);
  if (nmo > 0) std::fill(Am, Am + nmo * lda_m, 0.0);
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j <= i; ++j) for (int64_t k = 0; k < o; ++k) for (int64_t l = 0; l <= k; ++l) {
    Ap[(i * (i + 1) / 2 + j) * lda_p + k * (k + 1) / 2 + l] = (k == l) ? w(k, l, i, j) : w(k, l, i, j) + w(k, l, j, i);
    if (i > j && k > l) Am[(i * (i - 1) / 2 + j) * lda_m + k * (k - 1) / 2 + l] = w(k, l, i, j) - w(k, l, j, i);
  }
  return 0;
}
int dev_pack_w_pm_sum(int64_t o, const double* Wt, const double* X, const double* At, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) {
  std::vector<double> W((size_t)(o * o * o * o));
  for (int64_t k = 0; k < o; ++k) for (int64_t l = 0; l < o; ++l) for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j < o; ++j)
    W[(size_t)(((k * o + l) * o + i) * o + j)] = ((Wt[((i * o + j) * o + k) * o + l] + X[((i * o + j) * o + k) * o + l]) + At[((j * o + i) * o + k) * o + l]) + At[((i * o + j) * o + l) * o + k];
  return dev_pack_w_pm(o, W.data(), Ap, lda_p, Am, lda_m);
}
int dev_ccsd_t1_small(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, double* t1n) {
  for (int64_t i = 0; i < o; ++i) {
    std::vector<double> w((size_t)o);
    for (int64_t k = 0; k < o; ++k) { double q = 0; for (int64_t c = 0; c < v; ++c) q += t1[i * v + c] * Fov[k * v + c]; w[(size_t)k] = q - Loo[k * o + i]; }
    for (int64_t a = 0; a < v; ++a) {
      double s = 0;
      for (int64_t c = 0; c < v; ++c) s += t1[i * v + c] * Lvv[a * v + c];
      for (int64_t k = 0; k < o; ++k) s += w[(size_t)k] * t1[k * v + a];
      t1n[i * v + a] = s;
    }
  }
  return 0;
}
int dev_ccsd_t1_assemble(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, const double* S, const double* Lph1,
                         const double* PA, int SA, int64_t strideA, const double* PB, int SB, int64_t strideB, double* t1n) {
  const int64_t nov = o * v;
  std::vector<double> out((size_t)nov);
  int rc = dev_ccsd_t1_small(o, v, t1, Lvv, Loo, Fov, out.data());
  if (rc) return rc;
  for (int64_t r = 0; r < nov; ++r) {
    double s = out[(size_t)r];
    for (int64_t c = 0; c < nov; ++c) s += S[r * nov + c] * Fov[c] + Lph1[r * nov + c] * t1[c];
    for (int sl = 0; sl < SA; ++sl) s += PA[sl * strideA + r];
    for (int sl = 0; sl < SB; ++sl) s -= PB[sl * strideB + r];
    t1n[r] = s;
  }
  return 0;
}
int dev_ccsd_finish_t2_rings(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* RS, const double* M, const double* eo, const double* ev, double* t1n) {
  const int64_t ov = o * v;
  auto F = [&](int64_t i, int64_t j, int64_t a, int64_t b) {
    return U[((i * o + j) * v + a) * v + b] + RS[(i * v + a) * ov + j * v + b] - 0.5 * M[(i * v + a) * ov + j * v + b] - M[(i * v + b) * ov + j * v + a];
  };
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j <= i; ++j) for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b < v; ++b) {
    const int64_t idx = ((i * o + j) * v + a) * v + b;
    if (i == j && b > a) continue;      // the diagonal pair: each (a,b),(b,a) couple once, from the lower triangle's inputs
    const double r = (t2n[idx] + OV[idx] + F(i, j, a, b) + F(j, i, b, a)) / (eo[i] + eo[j] - ev[a] - ev[b]);
    t2n[idx] = r;
    t2n[((j * o + i) * v + b) * v + a] = r;
  }
  if (t1n) for (int64_t t = 0; t < ov; ++t) t1n[t] /= eo[t / v] - ev[t % v];
  return 0;
}
int dev_ccsd_ph_layouts(int64_t o, int64_t v, const double* t2, const double* t1, double* T, double* Tp, double* S, double* Ut, double* Tpt, double* Th) {
  for (int64_t k = 0; k < o; ++k) for (int64_t c = 0; c < v; ++c) for (int64_t j = 0; j < o; ++j) for (int64_t b = 0; b < v; ++b) {
    const int64_t off = ((k * v + c) * o + j) * v + b;
    const double x = t2[((k * o + j) * v + c) * v + b], xp = t2[((k * o + j) * v + b) * v + c], tt = 2.0 * t1[j * v + c] * t1[k * v + b];
    T[off] = x; Tp[off] = xp; S[off] = 2.0 * x - xp; Ut[off] = 2.0 * x - xp - tt; Tpt[off] = xp + tt;
    Th[((k * o + j) * v + c) * v + b] = 2.0 * xp - x;
  }
  return 0;
}
int dev_small_k_update(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sA, const double* B, int64_t sB,
                       double* C, int64_t sC) {
  for (int64_t z = 0; z < batch; ++z) for (int64_t m = 0; m < M; ++m) for (int64_t n = 0; n < N; ++n) {
    double s = 0.0;
    for (int64_t k = 0; k < K; ++k) s += A[z * sA + k * M + m] * B[z * sB + k * N + n];
    C[z * sC + m * N + n] += alpha * s;
  }
  return 0;
}
int dev_ccsd_y_traces(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add, int S, int64_t stride, double scale) {
  for (int64_t a = 0; a < v; ++a) for (int64_t c = 0; c < v; ++c) {
    double s = 0.0;
    for (int64_t k = 0; k < o; ++k) s += 2.0 * ZC[((k * o + k) * v + a) * v + c] - ZB[((k * v + c) * v + a) * o + k];
    if (add) { double t = 0.0; for (int sl = 0; sl < std::max(S, 1); ++sl) t += add[sl * stride + a * v + c]; s += scale * t; }
    Y[a * v + c] = s;
  }
  return 0;
}
int dev_gather_rows(int64_t nrows, int64_t len, const int64_t* idx, const double* src, int64_t ld, double* dst) {
  for (int64_t r = 0; r < nrows; ++r) for (int64_t c = 0; c < len; ++c) dst[r * len + c] = idx[r] < 0 ? 0.0 : src[idx[r] * ld + c];
  return 0;
}
int dev_scale_rows(int64_t nrows, int64_t len, double* x, const double* s) { for (int64_t r = 0; r < nrows; ++r) for (int64_t c = 0; c < len; ++c) x[r * len + c] *= s[r]; return 0; }
int dev_threshold_mask(int64_t n, const double* x, double eps, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = std::fabs(x[i]) >= eps ? 1.0 : 0.0; return 0; }
int dev_mul_bcast_rows(int64_t rows, int64_t cols, double* x, const double* m) { for (int64_t r = 0; r < rows; ++r) for (int64_t c = 0; c < cols; ++c) x[r * cols + c] *= m[c]; return 0; }
int dev_dot(int64_t n, const double* x, const double* y, double* o) { long double s = 0; for (int64_t i = 0; i < n; ++i) s += (long double)x[i] * y[i]; *o = (double)s; return 0; }
int dev_ccsd_finish_t2(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* eo, const double* ev) {
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j < o; ++j) for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b < v; ++b) {
    const int64_t idx = ((i * o + j) * v + a) * v + b;
    t2n[idx] = (t2n[idx] + OV[idx] + U[idx] + U[((j * o + i) * v + b) * v + a]) / (eo[i] + eo[j] - ev[a] - ev[b]);
  }
  return 0;
}
int dev_dot_many(int64_t n, const double* x, int m, const double* const* ys, double* o) {
  for (int j = 0; j < m; ++j) { double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i] * ys[j][i]; o[j] = s; }
  return 0;
}
int dev_batch_begin() { return 0; }      // (the mock executes every call at once: nothing to collect)
int dev_batch_flush() { return 0; }
int dev_wait_flag(const void* flag_host, unsigned long long seq) { if (*(const unsigned long long*)flag_host == seq) return 0; set_error("dev_wait_flag: word not written"); return QEMB_ERR_DEVICE; }
int dev_diis_push(int64_t n, const double* trial, const double* prev, double* e, double* xcopy, int m, const double* const* ys, int self, double* row_dev, double* row_host, void* flag_host, unsigned long long seq) {
  if (m <= 0 || m > 8 || self < 0 || self >= m) { set_error("dev_diis_push: 1 <= m <= 8 vectors, 0 <= self < m"); return QEMB_ERR_ARG; }
  for (int64_t i = 0; i < n; ++i) { const double t = trial[i]; e[i] = t - prev[i]; if (xcopy) xcopy[i] = t; }
  for (int j = 0; j < m; ++j) { const double* y = (j == self) ? e : ys[j]; double s = 0; for (int64_t i = 0; i < n; ++i) s += e[i] * y[i]; row_dev[j] = s; row_host[j] = s; }
  if (flag_host) *(unsigned long long*)flag_host = seq;
  return 0;
}
int dev_ccsd_extrapolate_energy(int64_t o, int64_t v, int nterms, const double* coef, const double* const* xs, double* amp, const double* L, double* tau, double* e_dev, double* e_host, void* flag_host, unsigned long long seq) {
  if (nterms <= 0 || nterms > 8) { set_error("dev_ccsd_extrapolate_energy: 1 <= nterms <= 8"); return QEMB_ERR_ARG; }
  const int64_t nov = o * v, na = nov + o * o * v * v;
  if (!(nterms == 1 && xs[0] == amp && coef[0] == 1.0)) {
    std::vector<double> x((size_t)na);
    for (int64_t t = 0; t < na; ++t) { double acc = 0.0; for (int q = 0; q < nterms; ++q) acc += coef[q] * xs[q][t]; x[(size_t)t] = acc; }
    std::copy(x.begin(), x.end(), amp);
  }
  long double s = 0;
  for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j < o; ++j) for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b < v; ++b) {
    const int64_t idx = ((i * o + j) * v + a) * v + b;
    tau[idx] = amp[nov + idx] + amp[i * v + a] * amp[j * v + b];
    s += (long double)L[idx] * tau[idx];
  }
  *e_dev = (double)s; *e_host = (double)s;
  if (flag_host) *(unsigned long long*)flag_host = seq;
  return 0;
}
int dev_absmax(int64_t n, const double* x, double* o) { double m = 0; for (int64_t i = 0; i < n; ++i) m = std::max(m, std::fabs(x[i])); *o = m; return 0; }
int dev_gemv_rows(int64_t rows, int64_t cols, const double* T, int64_t ldt, const double* x, double* y, double alpha, double beta) {
  for (int64_t r = 0; r < rows; ++r) { double s = 0; for (int64_t c = 0; c < cols; ++c) s += T[r * ldt + c] * x[c]; y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s; }
  return 0;
}
int dev_gemv_rows2(int64_t rows, int64_t cols, const double* T1, int64_t ld1, const double* x1, const double* T2, int64_t ld2, const double* x2, double* y, double alpha, double beta) {
  for (int64_t r = 0; r < rows; ++r) {
    double s = 0;
    for (int64_t c = 0; c < cols; ++c) s += T1[r * ld1 + c] * x1[c] + T2[r * ld2 + c] * x2[c];
    y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s;
  }
  return 0;
}
int dev_gemv_rows_two(int64_t rows1, int64_t cols1, const double* T1, int64_t ld1, const double* x1, double* y1, double a1, double b1,
                      int64_t rows2, int64_t cols2, const double* T2, int64_t ld2, const double* x2, double* y2, double a2, double b2) {
  int rc = dev_gemv_rows(rows1, cols1, T1, ld1, x1, y1, a1, b1);
  return rc ? rc : dev_gemv_rows(rows2, cols2, T2, ld2, x2, y2, a2, b2);
}
int dev_gemv_rows_batched(int64_t rows, int64_t cols, int64_t nbatch, const double* T, int64_t ldt, int64_t strideT, const double* x,
                          int64_t stridex, double* y, double alpha, double beta) {
  for (int64_t r = 0; r < rows; ++r) {
    double s = 0;
    for (int64_t b = 0; b < nbatch; ++b) for (int64_t c = 0; c < cols; ++c) s += T[b * strideT + r * ldt + c] * x[b * stridex + c];
    y[r] = (beta != 0.0) ? alpha * s + beta * y[r] : alpha * s;
  }
  return 0;
}
int dev_contract_mid(int64_t outer, int64_t mid, int64_t inner, const double* T, const double* x, double* Y, int64_t ldy, double alpha, double beta) {
  std::vector<double> acc((size_t)inner);
  for (int64_t p = 0; p < outer; ++p) {
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int64_t m = 0; m < mid; ++m) { const double xm = x[m]; const double* t = T + (p * mid + m) * inner; for (int64_t r = 0; r < inner; ++r) acc[r] += xm * t[r]; }
    for (int64_t r = 0; r < inner; ++r) { double* y = Y + p * ldy + r; *y = (beta != 0.0) ? alpha * acc[r] + beta * (*y) : alpha * acc[r]; }
  }
  return 0;
}
int dev_unpack_s4(int64_t n, const double* s4, double* s1) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < n; ++j) for (int64_t k = 0; k < n; ++k) for (int64_t l = 0; l < n; ++l)
    s1[((i * n + j) * n + k) * n + l] = s4[pidx(i, j) * np + pidx(k, l)];
  return 0;
}
int dev_pack_s4(int64_t n, const double* s1, double* s4) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j <= i; ++j) for (int64_t k = 0; k < n; ++k) for (int64_t l = 0; l <= k; ++l)
    s4[pidx(i, j) * np + pidx(k, l)] = s1[((i * n + j) * n + k) * n + l];
  return 0;
}
int dev_unpack_s8_to_s4(int64_t n, const double* s8, double* s4) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t r = 0; r < np; ++r) for (int64_t c = 0; c < np; ++c) s4[r * np + c] = s8[pidx(r, c)];
  return 0;
}
int dev_unpack_tril_rows_ld(int64_t rows, int64_t n, int64_t ld, const double* p, double* f) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t r = 0; r < rows; ++r) for (int64_t k = 0; k < n; ++k) for (int64_t l = 0; l < n; ++l) f[(r * n + k) * ld + l] = p[r * np + pidx(k, l)];
  return 0;
}
int dev_unpack_tril_rows(int64_t rows, int64_t n, const double* p, double* f) { return dev_unpack_tril_rows_ld(rows, n, n, p, f); }
int dev_unpack_tril_pair_rows_ld(int64_t nr, int64_t n, int64_t ld, const double* in, double* f) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t x = 0; x < nr; ++x) for (int64_t y = 0; y <= x; ++y) for (int64_t k = 0; k < n; ++k) for (int64_t l = 0; l < n; ++l)
    f[(pidx(x, y) * n + k) * ld + l] = in[(x * nr + y) * np + pidx(k, l)];
  return 0;
}
int dev_unpack_tril_pair_rows(int64_t nr, int64_t n, const double* in, double* f) { return dev_unpack_tril_pair_rows_ld(nr, n, n, in, f); }
int dev_pack_tril_rows(int64_t rows, int64_t n, const double* f, double* p) {
  const int64_t np = n * (n + 1) / 2;
  for (int64_t r = 0; r < rows; ++r) for (int64_t k = 0; k < n; ++k) for (int64_t l = 0; l <= k; ++l) p[r * np + pidx(k, l)] = f[(r * n + k) * n + l];
  return 0;
}

// cyclic two-sided Jacobi (independent of the product's one-sided formulation)
int dev_jacobi_eigh(int64_t n64, double* A, double* w, double* V, int* sweeps_out);
int dev_jacobi_eigh_until(int64_t n64, double* A, double* w, double* V, int* sweeps_out, double) { return dev_jacobi_eigh(n64, A, w, V, sweeps_out); }
int dev_jacobi_eigh(int64_t n64, double* A, double* w, double* V, int* sweeps_out) {
  const int n = (int)n64;
  std::vector<double> a(A, A + (size_t)n * n), v((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  int sweep = 0;
  for (; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off <= 1e-30 * std::max(diag, 1e-300)) break;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
      const double apq = a[(size_t)p * n + q];
      if (std::fabs(apq) < 1e-300) continue;
      const double th = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2 * apq);
      const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(1 + th * th));
      const double c = 1 / std::sqrt(1 + t * t), s = c * t;
      for (int k = 0; k < n; ++k) { const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q]; a[(size_t)k * n + p] = c * x - s * y; a[(size_t)k * n + q] = s * x + c * y; }
      for (int k = 0; k < n; ++k) { const double x = a[(size_t)p * n + k], y = a[(size_t)q * n + k]; a[(size_t)p * n + k] = c * x - s * y; a[(size_t)q * n + k] = s * x + c * y; }
      for (int k = 0; k < n; ++k) { const double x = v[(size_t)k * n + p], y = v[(size_t)k * n + q]; v[(size_t)k * n + p] = c * x - s * y; v[(size_t)k * n + q] = s * x + c * y; }
    }
  }
  std::vector<int> perm(n); std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return a[(size_t)x * n + x] < a[(size_t)y * n + y]; });
  for (int i = 0; i < n; ++i) { w[i] = a[(size_t)perm[i] * n + perm[i]]; for (int k = 0; k < n; ++k) V[(size_t)k * n + i] = v[(size_t)k * n + perm[i]]; }
  if (sweeps_out) *sweeps_out = sweep;
  return 0;
}
// fused steps of the fragment RHF of small fragments (linalg_f64.hip), restated with plain loops
int dev_scf_fused_max() { return 80; }
int dev_jacobi_eigh_in_basis(int64_t n64, const double* F, const double* Cp, double* w, double* C_out, double* C2_out, int nocc, double* dm_out, double, int* status_dev) {
  const int n = (int)n64;
  std::vector<double> a((size_t)n * n), v((size_t)n * n), c((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double t = 0;
    if (Cp) { for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) t += Cp[(size_t)k * n + i] * F[(size_t)k * n + l] * Cp[(size_t)l * n + j]; }
    else t = F[(size_t)i * n + j];
    a[(size_t)i * n + j] = t;
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) a[(size_t)i * n + j] = a[(size_t)j * n + i];
  int sweeps = 0;
  dev_jacobi_eigh(n, a.data(), w, v.data(), &sweeps);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double t = 0;
    if (Cp) { for (int k = 0; k < n; ++k) t += Cp[(size_t)i * n + k] * v[(size_t)k * n + j]; } else t = v[(size_t)i * n + j];
    c[(size_t)i * n + j] = t;
  }
  for (size_t t = 0; t < (size_t)n * n; ++t) { C_out[t] = c[t]; if (C2_out) C2_out[t] = c[t]; }
  if (dm_out) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int k = 0; k < nocc; ++k) t += c[(size_t)i * n + k] * c[(size_t)j * n + k]; dm_out[(size_t)i * n + j] = 2.0 * t; }
  *status_dev = sweeps;
  return 0;
}
int dev_scf_fock_small(int64_t n64, const double* h, const double* J, const double* K, const double* D, double* F, double* err, double* scal2) {
  const int n = (int)n64;
  long double e2 = 0, g2 = 0;
  for (size_t t = 0; t < (size_t)n * n; ++t) { F[t] = h[t] + J[t] - 0.5 * K[t]; e2 += (long double)(h[t] + F[t]) * D[t]; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double a = 0;
    for (int k = 0; k < n; ++k) a += F[(size_t)i * n + k] * D[(size_t)k * n + j] - D[(size_t)i * n + k] * F[(size_t)k * n + j];
    err[(size_t)i * n + j] = a; g2 += (long double)a * a;
  }
  scal2[0] = (double)e2; scal2[1] = (double)g2;
  return 0;
}
int dev_pack_density_sym(int64_t n64, const double* D, double* Dp) {
  const int64_t n = n64;
  for (int64_t r = 0; r < n; ++r) for (int64_t c = 0; c <= r; ++c) Dp[pidx(r, c)] = (r == c) ? D[r * n + r] : D[r * n + c] + D[c * n + r];
  return 0;
}
int dev_jacobi_svd(int64_t m64, int64_t n64, double* G, double* s, double* U, double* V, int* sweeps_out) {
  // via eigh of G^T G (adequate for a mock; the product does a genuine one-sided Jacobi)
  const int m = (int)m64, n = (int)n64;
  std::vector<double> gtg((size_t)n * n, 0.0), w(n), vv((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int k = 0; k < m; ++k) t += G[(size_t)k * n + i] * G[(size_t)k * n + j]; gtg[(size_t)i * n + j] = t; }
  dev_jacobi_eigh(n, gtg.data(), w.data(), vv.data(), sweeps_out);
  for (int i = 0; i < n; ++i) {           // descending
    const int src = n - 1 - i;
    s[i] = std::sqrt(std::max(w[src], 0.0));
    for (int k = 0; k < n; ++k) if (V) V[(size_t)k * n + i] = vv[(size_t)k * n + src];
    for (int k = 0; k < m; ++k) { double t = 0; for (int j = 0; j < n; ++j) t += G[(size_t)k * n + j] * vv[(size_t)j * n + src]; if (U) U[(size_t)k * n + i] = (s[i] > 1e-150) ? t / s[i] : 0.0; }
  }
  return 0;
}
int dev_cholesky_lower(int64_t n64, double* A) {
  const int n = (int)n64;
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0)) { set_error("Cholesky: matrix is not positive definite"); return QEMB_ERR_NUMERIC; }
    A[(size_t)j * n + j] = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) { double t = A[(size_t)i * n + j]; for (int k = 0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k]; A[(size_t)i * n + j] = t / A[(size_t)j * n + j]; }
    for (int k = j + 1; k < n; ++k) A[(size_t)j * n + k] = 0.0;
  }
  return 0;
}
int dev_tri_inverse_lower(int64_t n64, const double* L, double* X) {
  const int n = (int)n64;
  std::fill(X, X + (size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    X[(size_t)j * n + j] = 1.0 / L[(size_t)j * n + j];
    for (int i = j + 1; i < n; ++i) { double t = 0; for (int k = j; k < i; ++k) t += L[(size_t)i * n + k] * X[(size_t)k * n + j]; X[(size_t)i * n + j] = -t / L[(size_t)i * n + i]; }
  }
  return 0;
}

int dev_abs_overlap_prim(int nsh, const int* l, const double* ex, const double* xyz, const int64_t* cart0, int64_t ncart, int nroots,
                         const double* roots, const double* weights, double* out) {
  for (int i = 0; i < nsh; ++i) for (int j = 0; j <= i; ++j) {
    const int li = l[i], lj = l[j];
    const double ai = ex[i], aj = ex[j], aij = ai + aj, scale = 1.0 / std::sqrt(aij);
    double I[3][5][5] = {}, r2 = 0.0;
    for (int d = 0; d < 3; ++d) {
      const double Ra = xyz[3 * i + d], Rb = xyz[3 * j + d], Rp = (ai * Ra + aj * Rb) / aij;
      r2 += (Ra - Rb) * (Ra - Rb);
      for (int n = 0; n < nroots; ++n) {
        const double x = roots[n] * scale + Rp, xa = std::fabs(x - Ra), xb = std::fabs(x - Rb);
        double pa = 1.0;
        for (int p = 0; p <= li; ++p) { double pb = pa * weights[n]; for (int q = 0; q <= lj; ++q) { I[d][p][q] += pb; pb *= xb; } pa *= xa; }
      }
    }
    const double fac = scale * scale * scale * std::exp(-(ai * aj / aij) * r2);
    int ci = 0;
    for (int ix = li; ix >= 0; --ix) for (int iy = li - ix; iy >= 0; --iy) {
      const int iz = li - ix - iy;
      int cj = 0;
      for (int jx = lj; jx >= 0; --jx) for (int jy = lj - jx; jy >= 0; --jy) {
        const int jz = lj - jx - jy;
        const double v = I[0][ix][jx] * I[1][iy][jy] * I[2][iz][jz] * fac;
        out[(cart0[i] + ci) * ncart + cart0[j] + cj] = v;
        out[(cart0[j] + cj) * ncart + cart0[i] + ci] = v;
        ++cj;
      }
      ++ci;
    }
  }
  return 0;
}

}  // namespace qemb
