// api.cpp -- extern "C" surface of libqemb_hip.so (declared in include/qemb_hip.h).
// Thin argument marshalling only; the work is in the drivers (ccsd.cpp, scf.cpp, ao2mo.cpp, schmidt.cpp)
// and the device layer (dev_ops.h).
#include <cstring>
#include <cstdlib>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "../../include/qemb_hip_ops.h"
#include "dev_ops.h"
#include "ccsd.h"
#include "fragment.h"
#include "ao2mo.h"

using namespace qemb;
namespace qemb {
int dev_mfma_f64_peak(int iters, int blocks_per_cu, double* tflops);
int schmidt_eigh(const double* lmo, int N, int nmo, int nocc, const int64_t* frag, int n_f, double thr, double* TA_out,
                 int ld_out, int* n_b_out, int* sweeps_out);
int schmidt_subspace(const double* lmo, int N, int nmo, int nocc, const int64_t* frag, int n_f, double thr, double* TA_out,
                     int ld_out, int* n_b_out, int* sweeps_out);
int schmidt_svd(const double* rdm, int N, const int64_t* frag_in, int n_f, double thr, double* TA_out, int ld_out, int* n_b_out,
                int* sweeps_out);
int nsocc_guess(const double* Cproj, int n, int nocc, double* P_out, int* nsocc, double* mo_out);
}

static thread_local int g_test_ksplit = 0;   // qemb_set_gemm_ksplit: split-K override of qemb_op_gemm (tests / tuning), per calling thread

extern "C" {

// QEMB_BACKTRACE=1: native frames to stderr on SIGSEGV / SIGABRT (a debugging aid: where inside the library did a host fault happen?)
static void qemb_fault_handler(int sig) {
  void* frames[64];
  const int nfr = backtrace(frames, 64);
  const char msg[] = "[qemb] fatal signal; native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, nfr, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
int qemb_init(int device) {
  static const bool bt = [] { const char* e = std::getenv("QEMB_BACKTRACE"); if (e && e[0] != '0') { signal(SIGSEGV, qemb_fault_handler); signal(SIGABRT, qemb_fault_handler); } return true; }();
  (void)bt;
  return dev_init(device);
}
const char* qemb_last_error(void) { return last_error(); }
const char* qemb_backend(void) { return dev_backend_name(); }
int qemb_sync(void) { return dev_sync(); }
int qemb_device_sync(void) { return dev_sync_device(); }
int qemb_mem_info(size_t* f, size_t* t) { return dev_mem_info(f, t); }
int qemb_malloc(void** p, size_t bytes) { return dev_alloc(p, bytes); }
int qemb_free(void* p) { return dev_free(p); }
int qemb_trim(void) { return dev_trim(); }
int qemb_trim_all(void) { return dev_trim_all(); }
int qemb_h2d(void* d, const void* h, size_t b) { return dev_h2d(d, h, b); }
int qemb_d2h(void* h, const void* d, size_t b) { return dev_d2h(h, d, b); }
int qemb_h2d_async(void* d, const void* h, size_t b) { return dev_h2d_async(d, h, b); }
int qemb_d2d(void* d, const void* s, size_t b) { return dev_d2d(d, s, b); }
int qemb_timer_begin(int s) { return dev_timer_begin(s); }
int qemb_timer_end(int s) { return dev_timer_end(s); }
int qemb_timer_read(int s, double* ms, int64_t* c) { return dev_timer_read(s, ms, c); }
int qemb_timer_reset(int s) { return dev_timer_reset(s); }
int qemb_timer_live_events(int s) { return dev_timer_live_events(s); }

int qemb_op_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int a_kcontig,
                 int64_t strideA, const double* B, int64_t ldb, int b_kcontig, int64_t strideB, double beta,
                 double* C, int64_t ldc, int64_t strideC, int64_t batch) {
  GemmDesc g{M, N, K, alpha, beta, A, lda, a_kcontig, strideA, B, ldb, b_kcontig, strideB, C, ldc, strideC, batch, -1, g_test_ksplit};
  return dev_gemm(g);
}
int qemb_op_gemm_probe(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, int a_kcontig, const double* B, int64_t ldb, int b_kcontig,
                       double* C, int64_t ldc, int cfg, int ksplit, double* ms, double* clock_ghz, int64_t* workgroups) {
  GemmDesc g{M, N, K, 1.0, 0.0, A, lda, a_kcontig, 0, B, ldb, b_kcontig, 0, C, ldc, 0, 1, cfg, ksplit};
  long long wg = 0;
  const int rc = dev_gemm_probe(g, ms, clock_ghz, &wg);
  if (workgroups) *workgroups = wg;
  return rc;
}
int qemb_op_gemm_stamps(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int cfg, int ksplit,
                        double* out7) {
  GemmDesc g{M, N, K, 1.0, 0.0, A, lda, 1, 0, B, ldb, 1, 0, C, ldc, 0, 1, cfg, ksplit};
  return dev_gemm_stamps(g, 8, out7);
}
int qemb_set_gemm_ksplit(int ks) { g_test_ksplit = ks; return QEMB_OK; }
int qemb_set_gemm_config(int cfg) { dev_gemm_set_force_cfg(cfg); return QEMB_OK; }
#ifndef QEMB_HOSTCHECK
int qemb_mfma_f64_peak(int iters, int blocks_per_cu, double* tflops) { return dev_mfma_f64_peak(iters, blocks_per_cu, tflops); }
#else
int qemb_mfma_f64_peak(int, int, double*) { set_error("not available in the hostcheck build"); return QEMB_ERR_DEVICE; }
#endif
int qemb_pair_gemm_choice(int64_t rows, int64_t cols, int* cfg, int* ksplit) {
  int c = -1, k = 0;
  pick_pair_gemm(rows, cols, c, k);
  if (cfg) *cfg = c;
  if (ksplit) *ksplit = k;
  return QEMB_OK;
}
int qemb_set_gemm_splitk(int enabled) { dev_gemm_set_auto_splitk(enabled); return QEMB_OK; }
int qemb_op_copy4(const int64_t dim[4], const double* in, const int64_t si[4], double* out, const int64_t so[4],
                  double alpha, double beta) {
  Copy4Desc c{};
  for (int k = 0; k < 4; ++k) { c.dim[k] = dim[k]; c.si[k] = si[k]; c.so[k] = so[k]; }
  c.in = in; c.out = out; c.alpha = alpha; c.beta = beta;
  return dev_copy4(c);
}
int qemb_op_outer4(const int64_t dim[4], const double* u, int64_t su0, int64_t su2, const double* v, int64_t sv1,
                   int64_t sv3, double* out, const int64_t so[4], double alpha, double beta) {
  Outer4Desc o{};
  for (int k = 0; k < 4; ++k) { o.dim[k] = dim[k]; o.so[k] = so[k]; }
  o.u = u; o.su0 = su0; o.su2 = su2; o.v = v; o.sv1 = sv1; o.sv3 = sv3; o.out = out; o.alpha = alpha; o.beta = beta;
  return dev_outer4(o);
}
int qemb_op_div_denom(double* x, int64_t d0, int64_t d1, int64_t d2, int64_t d3, const double* ea, const double* eb,
                      const double* ec, const double* ed) { return dev_div_denom(x, d0, d1, d2, d3, ea, eb, ec, ed); }
int qemb_op_ladder_pack_vvvv(int64_t n, int64_t o, const double* M, double* Vp, int64_t ldp, double* Vm, int64_t ldm) { return dev_ladder_pack_vvvv(n, o, M, Vp, ldp, Vm, ldm); }
int qemb_op_ladder_pack_tau(int64_t o, int64_t v, const double* tau, double* Tp, int64_t ldp, double* Tm, int64_t ldm) { return dev_ladder_pack_tau(o, v, tau, Tp, ldp, Tm, ldm); }
int qemb_op_ladder_scatter_pm(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, double* t2) { return dev_ladder_scatter_pm(o, v, Rp, ldp, Rm, ldm, t2); }
int qemb_op_dot(int64_t n, const double* x, const double* y, double* o) { return dev_dot(n, x, y, o); }
int qemb_op_absmax(int64_t n, const double* x, double* o) { return dev_absmax(n, x, o); }
int qemb_op_gemv_rows(int64_t rows, int64_t cols, const double* T, int64_t ldt, const double* x, double* y, double alpha,
                      double beta) { return dev_gemv_rows(rows, cols, T, ldt, x, y, alpha, beta); }
int qemb_op_gemv_rows_batched(int64_t rows, int64_t cols, int64_t nbatch, const double* T, int64_t ldt, int64_t strideT, const double* x,
                              int64_t stridex, double* y, double alpha, double beta) { return dev_gemv_rows_batched(rows, cols, nbatch, T, ldt, strideT, x, stridex, y, alpha, beta); }
int qemb_op_contract_mid(int64_t outer, int64_t mid, int64_t inner, const double* T, const double* x, double* Y,
                         int64_t ldy, double alpha, double beta) { return dev_contract_mid(outer, mid, inner, T, x, Y, ldy, alpha, beta); }
int qemb_op_unpack_s4(int64_t n, const double* s4, double* s1) { return dev_unpack_s4(n, s4, s1); }
int qemb_op_pack_s4(int64_t n, const double* s1, double* s4) { return dev_pack_s4(n, s1, s4); }
int qemb_op_unpack_s8_to_s4(int64_t n, const double* s8, double* s4) { return dev_unpack_s8_to_s4(n, s8, s4); }
int qemb_comm_unique_id(void* id) { return dev_comm_unique_id(id); }
int qemb_comm_init(int rank, int world, const void* id) { return dev_comm_init(rank, world, id); }
int qemb_comm_info(int* rank, int* world) { return dev_comm_info(rank, world); }
int qemb_comm_allreduce(double* buf, int64_t n, int op) { return dev_comm_allreduce(buf, n, op); }
int qemb_comm_destroy(void) { return dev_comm_destroy(); }
int qemb_ctx_count(int n) { return dev_ctx_count(n); }
int qemb_ctx_bind(int k) { return dev_ctx_bind(k); }
int qemb_ctx_partition(int parts) { return dev_ctx_partition(parts); }
int qemb_op_gemm_slab_rows(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, int64_t a_slab, int64_t a_slab_skip, const double* B, int64_t ldb,
                           int b_kcontig, double* C, int64_t ldc, int cfg) {
  GemmDesc g{};
  g.M = M; g.N = N; g.K = K; g.alpha = 1.0; g.beta = 0.0;
  g.A = A; g.lda = lda; g.a_kcontig = 0; g.strideA = 0; g.a_slab = a_slab; g.a_slab_skip = a_slab_skip;
  g.B = B; g.ldb = ldb; g.b_kcontig = b_kcontig; g.strideB = 0;
  g.C = C; g.ldc = ldc; g.strideC = 0; g.batch = 1; g.cfg = cfg; g.ksplit = 0;
  return dev_gemm(g);
}
int qemb_gemm_flop_count(double* flops, int reset) { return dev_gemm_flop_count(flops, reset); }
int qemb_tape_cache_counters(int64_t* reused, int64_t* recorded, int reset) {
  long long a = 0, b = 0;
  tape_cache_counters(&a, &b, reset);
  if (reused) *reused = a;
  if (recorded) *recorded = b;
  return QEMB_OK;
}
int qemb_alloc_stats(long long* n, long long* nfree, double* ms, double* gb, int reset) { return dev_alloc_stats(n, nfree, ms, gb, reset); }
int qemb_ctx_timer_read(int ctx, int slot, double* total_ms, int64_t* count, int reset) { return dev_ctx_timer_read(ctx, slot, total_ms, count, reset); }
int qemb_op_k_from_pairs(int64_t n, const double* H, const double* D, double* K) { return dev_k_from_pairs(n, H, D, K); }
int qemb_op_jk_from_packed(int64_t n, const double* S4, const double* D, const double* Dp, double* Jp, double* K) { return dev_jk_from_packed(n, S4, D, Dp, Jp, K); }
int qemb_op_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm) { return dev_pack_pm_cols(rows, v, in, Op, ldp, Om, ldm); }
int qemb_op_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out) { return dev_scatter_pm_rows(o, ncols, Xp, Xm, out); }
int qemb_op_pack_w_pm(int64_t o, const double* W, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) { return dev_pack_w_pm(o, W, Ap, lda_p, Am, lda_m); }
int qemb_op_ladder_scatter_pm2(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, const double* Hp, const double* Hm,
                               int assign, double* t2) { return dev_ladder_scatter_pm2(o, v, Rp, ldp, Rm, ldm, Hp, Hm, assign, t2); }
int qemb_op_lincomb2(int64_t n, double a, const double* x, double b, const double* y, double beta, double* out) {
  const double c[2] = {a, b};
  const double* xs[2] = {x, y};
  return dev_lincomb(n, 2, c, xs, beta, out);
}
int qemb_op_mirror_lower(int64_t n, double* A, int64_t lda) { return dev_mirror_lower(n, A, lda); }
int qemb_op_small_k_update(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sA, const double* B, int64_t sB, double* C, int64_t sC) {
  return dev_small_k_update(batch, M, N, K, alpha, A, sA, B, sB, C, sC);
}
int qemb_op_ccsd_ph_layouts(int64_t o, int64_t v, const double* t2, const double* t1, double* T, double* Tp, double* S, double* Ut, double* Tpt, double* Th) {
  return dev_ccsd_ph_layouts(o, v, t2, t1, T, Tp, S, Ut, Tpt, Th);
}
int qemb_op_copy4_two(const int64_t dim[4], const double* in, const int64_t si[4], double* out, const int64_t so[4], double alpha, double beta, const double* base,
                      double* out2, const double* in2, double c2a, double c2b) {
  Copy4Desc c{};
  for (int k = 0; k < 4; ++k) { c.dim[k] = dim[k]; c.si[k] = si[k]; c.so[k] = so[k]; }
  c.in = in; c.out = out; c.alpha = alpha; c.beta = beta; c.base = base;
  c.out2 = out2; c.in2 = in2; c.c2a = c2a; c.c2b = c2b;
  return dev_copy4(c);
}
int qemb_op_scatter_pm_rows_add(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out, const double* add) { return dev_scatter_pm_rows(o, ncols, Xp, Xm, out, add); }
int qemb_op_ccsd_y_traces_add(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add) { return dev_ccsd_y_traces(o, v, ZC, ZB, Y, add); }
int qemb_op_pack_w_pm_sum(int64_t o, const double* Wt, const double* X, const double* At, double* Ap, int64_t lda_p, double* Am, int64_t lda_m) {
  return dev_pack_w_pm_sum(o, Wt, X, At, Ap, lda_p, Am, lda_m);
}
int qemb_op_ccsd_t1_small(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, double* t1n) { return dev_ccsd_t1_small(o, v, t1, Lvv, Loo, Fov, t1n); }
int qemb_op_ccsd_t1_assemble(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, const double* S, const double* Lph1,
                             const double* PA, int SA, int64_t strideA, const double* PB, int SB, int64_t strideB, double* t1n) {
  return dev_ccsd_t1_assemble(o, v, t1, Lvv, Loo, Fov, S, Lph1, PA, SA, strideA, PB, SB, strideB, t1n);
}
int qemb_op_gemv_rows_two(int64_t rows1, int64_t cols1, const double* T1, int64_t ld1, const double* x1, double* y1, double a1, double b1,
                          int64_t rows2, int64_t cols2, const double* T2, int64_t ld2, const double* x2, double* y2, double a2, double b2) {
  return dev_gemv_rows_two(rows1, cols1, T1, ld1, x1, y1, a1, b1, rows2, cols2, T2, ld2, x2, y2, a2, b2);
}
int qemb_op_ccsd_y_traces_slabs(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add, int S, int64_t stride, double scale) {
  return dev_ccsd_y_traces(o, v, ZC, ZB, Y, add, S, stride, scale);
}
int qemb_op_gemv_rows2(int64_t rows, int64_t cols, const double* T1, int64_t ld1, const double* x1, const double* T2, int64_t ld2, const double* x2, double* y, double alpha, double beta) {
  return dev_gemv_rows2(rows, cols, T1, ld1, x1, T2, ld2, x2, y, alpha, beta);
}
int qemb_op_ccsd_finish_t2_rings(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* RS, const double* M, const double* eo, const double* ev, double* t1n) {
  return dev_ccsd_finish_t2_rings(o, v, t2n, U, OV, RS, M, eo, ev, t1n);
}
int qemb_op_diis_push(int64_t n, const double* trial, const double* prev, double* e, double* xcopy, int m, const double* const* ys, int self, double* row_dev, double* row_host) {
  void* pin = nullptr;
  QTRY(dev_pinned_alloc(&pin, sizeof(double) * 8));
  int rc = dev_diis_push(n, trial, prev, e, xcopy, m, ys, self, row_dev, (double*)pin);
  if (!rc) rc = dev_sync();
  if (!rc) for (int j = 0; j < m; ++j) row_host[j] = ((double*)pin)[j];
  dev_pinned_free(pin);
  return rc;
}
int qemb_op_ccsd_extrapolate_energy(int64_t o, int64_t v, int nterms, const double* coef, const double* const* xs, double* amp, const double* L, double* tau, double* e_host) {
  void* pin = nullptr;
  DBuf e_dev;
  QTRY(e_dev.alloc(1));
  QTRY(dev_pinned_alloc(&pin, sizeof(double) * 8));
  int rc = dev_ccsd_extrapolate_energy(o, v, nterms, coef, xs, amp, L, tau, e_dev, (double*)pin);
  if (!rc) rc = dev_sync();
  if (!rc) *e_host = *(double*)pin;
  dev_pinned_free(pin);
  return rc;
}
int qemb_op_ccsd_y_traces(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y) { return dev_ccsd_y_traces(o, v, ZC, ZB, Y); }
int qemb_op_gather_rows(int64_t nrows, int64_t len, const int64_t* idx_dev, const double* src, int64_t ld, double* dst) { return dev_gather_rows(nrows, len, idx_dev, src, ld, dst); }
int qemb_op_scale_rows(int64_t nrows, int64_t len, double* x, const double* s) { return dev_scale_rows(nrows, len, x, s); }
int qemb_op_pack_pair_rows(int64_t n, int64_t ncols, const double* in, double* out) { return dev_pack_pair_rows(n, ncols, in, out); }
int qemb_op_extract_pf(int64_t n, const double* Mp, int64_t p0, int64_t q0, int64_t r0, int64_t s0, int64_t sp, int64_t sq, int64_t sr, int64_t ss, double* out) {
  return dev_extract_pf(n, Mp, p0, q0, r0, s0, sp, sq, sr, ss, out);
}
int qemb_op_extract_pf_t(int64_t n, const double* T, int64_t x0, int64_t r0, int64_t s0, int64_t c0, int64_t sx, int64_t sr, int64_t ss, int64_t sc, double* out) {
  return dev_extract_pf_t(n, T, x0, r0, s0, c0, sx, sr, ss, sc, out);
}
int qemb_op_ladder_pack_vvvv_pf(int64_t n, int64_t o, const double* Mp, double* Vp, int64_t ldp, double* Vm, int64_t ldm) { return dev_ladder_pack_vvvv_pf(n, o, Mp, Vp, ldp, Vm, ldm); }
int qemb_op_unpack_tril_pair_rows(int64_t nr, int64_t n, const double* in, double* full) { return dev_unpack_tril_pair_rows(nr, n, in, full); }
int qemb_op_unpack_tril_rows(int64_t rows, int64_t n, const double* p, double* f) { return dev_unpack_tril_rows(rows, n, p, f); }
int qemb_op_pack_tril_rows(int64_t rows, int64_t n, const double* f, double* p) { return dev_pack_tril_rows(rows, n, f, p); }
int qemb_op_jacobi_eigh(int64_t n, double* A, double* w, double* V, int* sweeps) { return dev_jacobi_eigh(n, A, w, V, sweeps); }
// the fused steps of the fragment RHF of small fragments (scf.cpp), each on its own
int qemb_op_scf_fused_max(void) { return dev_scf_fused_max(); }
int qemb_op_jacobi_eigh_in_basis(int64_t n, const double* F, const double* Cp, double* w, double* C_out, double* C2_out, int nocc, double* dm_out, double stop_below, int* sweeps) {
  DBuf st;
  QTRY(st.alloc(1));
  QTRY(dev_jacobi_eigh_in_basis(n, F, Cp, w, C_out, C2_out, nocc, dm_out, stop_below, reinterpret_cast<int*>(st.p)));
  double word = 0.0;
  QTRY(dev_d2h(&word, st.p, sizeof(double)));
  int sw; std::memcpy(&sw, &word, sizeof(int));
  if (sweeps) *sweeps = sw;
  if (sw < 0) { set_error("Jacobi sweeps did not converge in 40 sweeps"); return QEMB_ERR_NOCONV; }
  return QEMB_OK;
}
int qemb_op_scf_fock_small(int64_t n, const double* h, const double* J, const double* K, const double* D, double* F, double* err, double* scal2) { return dev_scf_fock_small(n, h, J, K, D, F, err, scal2); }
int qemb_op_pack_density_sym(int64_t n, const double* D, double* Dp) { return dev_pack_density_sym(n, D, Dp); }
int qemb_op_jacobi_svd(int64_t m, int64_t n, double* G, double* s, double* U, double* V, int* sweeps) { return dev_jacobi_svd(m, n, G, s, U, V, sweeps); }
int qemb_op_cholesky_lower(int64_t n, double* A) { return dev_cholesky_lower(n, A); }
int qemb_op_tri_inverse_lower(int64_t n, const double* L, double* Linv) { return dev_tri_inverse_lower(n, L, Linv); }

// ---------------------------------------------------------------- fragment solver ----------------
void qemb_default_opts(qemb_solver_opts* o) {
  CcsdOptions c; ScfOptions s;
  memset(o, 0, sizeof *o);
  o->struct_size = (uint32_t)sizeof(qemb_solver_opts);
  o->cc_conv_tol = c.conv_tol; o->cc_conv_tol_normt = c.conv_tol_normt; o->cc_max_cycle = c.max_cycle; o->cc_diis_space = c.diis_space;
  o->scf_conv_tol = s.conv_tol; o->scf_conv_tol_grad = s.conv_tol_grad; o->scf_max_cycle = s.max_cycle; o->scf_diis_space = s.diis_space;
  o->warm_start = 0; o->verbose = 0;
  LambdaOptions l;
  o->relax_density = 0; o->lambda_conv_tol = l.conv_tol; o->lambda_max_cycle = l.max_cycle;
  o->strict_convergence = 1;
}
static FragmentOptions to_opts(const qemb_solver_opts* o) {
  FragmentOptions f;
  if (o) {
    f.cc.conv_tol = o->cc_conv_tol; f.cc.conv_tol_normt = o->cc_conv_tol_normt; f.cc.max_cycle = o->cc_max_cycle;
    f.cc.diis_space = o->cc_diis_space; f.cc.verbose = o->verbose;
    f.scf.conv_tol = o->scf_conv_tol; f.scf.conv_tol_grad = o->scf_conv_tol_grad; f.scf.max_cycle = o->scf_max_cycle;
    f.scf.diis_space = o->scf_diis_space; f.scf.verbose = o->verbose;
    f.warm_start = o->warm_start;
    f.relax_density = o->relax_density; f.lam.conv_tol = o->lambda_conv_tol; f.lam.max_cycle = o->lambda_max_cycle;
    f.lam.diis_space = o->cc_diis_space; f.lam.verbose = o->verbose;
    f.strict = o->strict_convergence;
  }
  return f;
}
// NULL options = defaults; anything else must carry the size of THIS header's struct (set by qemb_default_opts)
#define CHECK_OPTS(o)                                                                                                     \
  do {                                                                                                                    \
    if ((o) && (o)->struct_size != (uint32_t)sizeof(qemb_solver_opts)) {                                                  \
      set_error("qemb_solver_opts.struct_size is " + std::to_string((o)->struct_size) + ", this library expects " +       \
                std::to_string(sizeof(qemb_solver_opts)) + ": initialise the struct with qemb_default_opts() (binding built " \
                "against another qemb_hip.h?)");                                                                          \
      return QEMB_ERR_ARG;                                                                                                \
    }                                                                                                                     \
  } while (0)
#define FRAG(f) (reinterpret_cast<Fragment*>(f))
#define CHECK_FRAG(f) do { if (!(f)) { set_error("null fragment handle"); return QEMB_ERR_ARG; } } while (0)

int qemb_frag_create(int n, int n_f, qemb_frag_t* out) {
  if (n <= 0 || n_f < 0 || n_f > n || !out) { set_error("qemb_frag_create: bad arguments"); return QEMB_ERR_ARG; }
  *out = new Fragment(n, n_f);
  return QEMB_OK;
}
int qemb_frag_free(qemb_frag_t f) { delete FRAG(f); return QEMB_OK; }
int qemb_frag_set_eri_s4(qemb_frag_t f, const double* s4) { CHECK_FRAG(f); return FRAG(f)->set_eri_s4_host(s4); }
int qemb_frag_set_eri_s4_dev(qemb_frag_t f, const double* s4) { CHECK_FRAG(f); return FRAG(f)->set_eri_s4_dev(s4); }
int qemb_frag_set_df_factor(qemb_frag_t f, int naux, const double* B) { CHECK_FRAG(f); return FRAG(f)->set_df_factor_host(naux, B); }
int qemb_frag_set_df_factor_dev(qemb_frag_t f, int naux, const double* B) { CHECK_FRAG(f); return FRAG(f)->set_df_factor_dev(naux, B); }
int qemb_frag_set_df_only(qemb_frag_t f, int naux, const double* B) { CHECK_FRAG(f); return FRAG(f)->set_df_only_host(naux, B); }
int qemb_frag_set_df_only_dev(qemb_frag_t f, int naux, const double* B) { CHECK_FRAG(f); return FRAG(f)->set_df_only_dev(naux, B); }
int qemb_frag_resident_bytes(qemb_frag_t f, int64_t* bytes) { CHECK_FRAG(f); if (!bytes) { set_error("qemb_frag_resident_bytes: null argument"); return QEMB_ERR_ARG; } *bytes = FRAG(f)->resident_bytes(); return QEMB_OK; }
int qemb_frag_mo_route(qemb_frag_t f, int route) { CHECK_FRAG(f); return FRAG(f)->set_mo_route(route); }
int qemb_frag_mo_route_used(qemb_frag_t f, int* used_factor, int* naux) {
  CHECK_FRAG(f);
  if (used_factor) *used_factor = FRAG(f)->last_route_was_factor() ? 1 : 0;
  if (naux) *naux = FRAG(f)->df_naux();
  return QEMB_OK;
}
int qemb_frag_get_eri_s4(qemb_frag_t f, double* s4) {
  CHECK_FRAG(f);
  if (!s4) { set_error("qemb_frag_get_eri_s4: null buffer"); return QEMB_ERR_ARG; }
  return FRAG(f)->export_eri_s4(s4);      // the resident block, or B^T B of a fragment that lives on its factor (formed for this call)
}
int qemb_frag_set_energy_data(qemb_frag_t f, const double* h1, const double* veff0, const double* veff, double weight,
                              const int* centers, int ncenter) {
  CHECK_FRAG(f);
  for (int i = 0; i < ncenter; ++i) if (centers[i] < 0 || centers[i] >= FRAG(f)->nf()) { set_error("centre index outside the fragment sites"); return QEMB_ERR_ARG; }
  FRAG(f)->set_energy_data(h1, veff0, veff, weight, centers, ncenter);
  return QEMB_OK;
}
int qemb_frag_jk(qemb_frag_t f, const double* P, double* J, double* K) { CHECK_FRAG(f); return FRAG(f)->hf_veff_from_dm(P, J, K); }
int qemb_frag_solve(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts, int eeval,
                    double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1, double* t2,
                    double* e_frag, double* e_corr_mo, double* e_scf, double* ebe_hf, int* n_iter, int* scf_cycles) {
  CHECK_FRAG(f); CHECK_OPTS(opts);
  if (!h) { set_error("qemb_frag_solve: h is NULL"); return QEMB_ERR_ARG; }
  FragmentResult r;
  int rc = FRAG(f)->solve(nsocc, h, dm0, to_opts(opts), eeval, &r, mo_coeff, mo_energy, rdm1_emb, rdm1_mo, t1, t2);
  if (n_iter) *n_iter = r.n_iter;
  if (scf_cycles) *scf_cycles = r.scf_cycles;
  FRAG(f)->last_lambda_iters = r.lambda_iters;
  if (rc < 0) return rc;
  if (e_frag) { e_frag[0] = r.e_frag[0]; e_frag[1] = r.e_frag[1]; e_frag[2] = r.e_frag[2]; }
  if (e_corr_mo) *e_corr_mo = r.e_corr_mo;
  if (e_scf) *e_scf = r.e_scf;
  if (ebe_hf) *ebe_hf = r.ebe_hf;
  return rc;        // QEMB_OK, or QEMB_WARN_NOCONV with strict_convergence = 0
}
int qemb_frag_solve_batch(int nfrag, const qemb_frag_t* frags, const int* nsocc, const double* const* h, const double* const* dm0,
                          const qemb_solver_opts* opts, int eeval, double* const* mo_coeff, double* const* mo_energy,
                          double* const* rdm1_emb, double* const* rdm1_mo, double* const* t1, double* const* t2, double* e_frag,
                          double* e_corr_mo, double* e_scf, double* ebe_hf, int* n_iter, int* scf_cycles, int64_t* stats) {
  if (nfrag < 0 || (nfrag > 0 && (!frags || !nsocc || !h))) { set_error("qemb_frag_solve_batch: bad arguments"); return QEMB_ERR_ARG; }
  CHECK_OPTS(opts);
  std::vector<Fragment*> frs; std::vector<int> o; std::vector<const double*> hs, dms; std::vector<Fragment::BatchOutputs> outs(nfrag);
  for (int f = 0; f < nfrag; ++f) {
    if (!frags[f] || !h[f]) { set_error("qemb_frag_solve_batch: null fragment handle or h"); return QEMB_ERR_ARG; }
    for (int g = 0; g < f; ++g) if (frags[g] == frags[f]) { set_error("qemb_frag_solve_batch: the same fragment twice"); return QEMB_ERR_ARG; }
    frs.push_back(FRAG(frags[f])); o.push_back(nsocc[f]); hs.push_back(h[f]); dms.push_back(dm0 ? dm0[f] : nullptr);
    auto pick = [&](double* const* arr) { return arr ? arr[f] : nullptr; };
    outs[f].mo_coeff = pick(mo_coeff); outs[f].mo_energy = pick(mo_energy); outs[f].rdm1_emb = pick(rdm1_emb);
    outs[f].rdm1_mo = pick(rdm1_mo); outs[f].t1 = pick(t1); outs[f].t2 = pick(t2);
  }
  std::vector<FragmentResult> res;
  LockstepStats st;
  const int rc = Fragment::solve_batch(frs, o, hs, dms, to_opts(opts), eeval, res, outs, &st);
  for (int f = 0; f < nfrag && f < (int)res.size(); ++f) {
    if (n_iter) n_iter[f] = res[f].n_iter;
    if (scf_cycles) scf_cycles[f] = res[f].scf_cycles;
    frs[f]->last_lambda_iters = res[f].lambda_iters;
    if (rc < 0) continue;
    if (e_frag) for (int k = 0; k < 3; ++k) e_frag[3 * f + k] = res[f].e_frag[k];
    if (e_corr_mo) e_corr_mo[f] = res[f].e_corr_mo;
    if (e_scf) e_scf[f] = res[f].e_scf;
    if (ebe_hf) ebe_hf[f] = res[f].ebe_hf;
  }
  if (stats) { stats[0] = st.merged_runs; stats[1] = st.launches; stats[2] = st.grouped; stats[3] = st.operations; stats[4] = st.max_group; }
  return rc;
}
int qemb_frag_lambda_iters(qemb_frag_t f, int* n_iter) { CHECK_FRAG(f); if (n_iter) *n_iter = FRAG(f)->last_lambda_iters; return QEMB_OK; }
int qemb_frag_scf(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts, double* mo_coeff,
                  double* mo_energy, double* J, double* K, double* e_scf, int* converged, int* cycles) {
  CHECK_FRAG(f); CHECK_OPTS(opts);
  if (!h) { set_error("qemb_frag_scf: h is NULL"); return QEMB_ERR_ARG; }
  ScfResult r;
  int rc = FRAG(f)->scf_only(nsocc, h, dm0, to_opts(opts).scf, mo_coeff, mo_energy, J, K, &r);
  if (rc) return rc;
  if (e_scf) *e_scf = r.e_tot;
  if (converged) *converged = r.converged ? 1 : 0;
  if (cycles) *cycles = r.cycles;
  return QEMB_OK;
}
int qemb_frag_cphf(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts, const double* vpots,
                   int npot, double* dPs) {
  CHECK_FRAG(f); CHECK_OPTS(opts);
  if (!h || !vpots || !dPs) { set_error("qemb_frag_cphf: null argument"); return QEMB_ERR_ARG; }
  return FRAG(f)->cphf_response(nsocc, h, dm0, to_opts(opts).scf, vpots, npot, dPs);
}
int qemb_ccsd_solve(int n, int nsocc, int n_f, const double* h, const double* eri_s4, const double* dm0,
                    const qemb_solver_opts* opts, const double* h1, const double* veff0, double weight, const int* centers,
                    int ncenter, double* mo_coeff, double* mo_energy, double* t1, double* t2, double* rdm1_emb, double* e_frag,
                    double* e_corr_mo, int* n_iter) {
  if (n <= 0 || n_f < 0 || n_f > n || !eri_s4) { set_error("qemb_ccsd_solve: bad arguments"); return QEMB_ERR_ARG; }
  CHECK_OPTS(opts);
  Fragment fr(n, n_f);
  int rc = fr.set_eri_s4_host(eri_s4);
  if (rc) return rc;
  const int eeval = (h1 && veff0 && e_frag) ? 1 : 0;
  if (eeval) fr.set_energy_data(h1, veff0, nullptr, weight, centers, ncenter);
  FragmentResult r;
  rc = fr.solve(nsocc, h, dm0, to_opts(opts), eeval, &r, mo_coeff, mo_energy, rdm1_emb, nullptr, t1, t2);
  if (n_iter) *n_iter = r.n_iter;
  if (rc < 0) return rc;
  if (e_frag && eeval) { e_frag[0] = r.e_frag[0]; e_frag[1] = r.e_frag[1]; e_frag[2] = r.e_frag[2]; }
  if (e_corr_mo) *e_corr_mo = r.e_corr_mo;
  return rc;
}
int qemb_frag_prepare_ccsd(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts) {
  CHECK_FRAG(f); CHECK_OPTS(opts); return FRAG(f)->prepare_ccsd(nsocc, h, dm0, to_opts(opts));
}
int qemb_frag_ccsd_iterate(qemb_frag_t f, int niter, double* e, double* nt) { CHECK_FRAG(f); return FRAG(f)->ccsd_iterate(niter, e, nt); }
int qemb_frag_ccsd_export(qemb_frag_t f, const char* name, double* host, int64_t nelem) { CHECK_FRAG(f); return FRAG(f)->ccsd_export(name, host, nelem); }
int qemb_frag_ccsd_reset(qemb_frag_t f) { CHECK_FRAG(f); return FRAG(f)->ccsd_reset(); }

// ---------------------------------------------------------------- ERI transforms ------------------
static int deliver_s4(DBuf& s4, int n, double* out_host, qemb_frag_t frag, DBuf* factor = nullptr, int naux = 0) {
  const int64_t np = (int64_t)n * (n + 1) / 2;
  if (out_host) { int rc = dev_d2h(out_host, s4, sizeof(double) * np * np); if (rc) return rc; }
  if (frag) {
    if (FRAG(frag)->n() != n) { set_error("fragment handle has a different n"); return QEMB_ERR_ARG; }
    int rc = FRAG(frag)->adopt_eri_s4(std::move(s4));    // the transform's result block becomes the fragment's resident ERIs (round 4: no 4.7 GB copy)
    if (rc == 0 && factor && factor->p) rc = FRAG(frag)->adopt_df_factor(std::move(*factor), naux);      // ... and the factor it was formed from goes with it
    return rc;
  }
  return QEMB_OK;
}
int qemb_aoeri_upload(int N, const double* eri, int sym, qemb_aoeri_t* out) {
  if (N <= 0 || !eri || !out) { set_error("qemb_aoeri_upload: bad arguments"); return QEMB_ERR_ARG; }
  AoEri* a = new AoEri();
  int rc = a->upload(N, eri, sym);
  if (rc) { delete a; return rc; }
  *out = a;
  return QEMB_OK;
}
int qemb_aoeri_free(qemb_aoeri_t ao) { delete reinterpret_cast<AoEri*>(ao); return QEMB_OK; }
int qemb_ao2mo_dense(qemb_aoeri_t ao, const double* TA, int n, double* out_s4_host, qemb_frag_t frag) {
  if (!ao || !TA) { set_error("qemb_ao2mo_dense: null argument"); return QEMB_ERR_ARG; }
  AoEri* a = reinterpret_cast<AoEri*>(ao);
  DBuf dTA, s4;
  int rc;
  if ((rc = dTA.alloc((int64_t)a->N * n))) return rc;
  if ((rc = dev_h2d(dTA, TA, sizeof(double) * a->N * n))) return rc;
  if ((rc = s4.alloc(((int64_t)n * (n + 1) / 2) * ((int64_t)n * (n + 1) / 2)))) return rc;
  if ((rc = ao2mo_dense(*a, dTA, n, s4))) return rc;
  return deliver_s4(s4, n, out_s4_host, frag);
}
int qemb_df_create(int naux, const double* j2c, qemb_df_t* out) {
  if (naux <= 0 || !j2c || !out) { set_error("qemb_df_create: bad arguments"); return QEMB_ERR_ARG; }
  DfContext* d = new DfContext();
  int rc = d->set_metric(naux, j2c);
  if (rc) { delete d; return rc; }
  *out = d;
  return QEMB_OK;
}
int qemb_lpq_upload(const double* L, int naux, qemb_df_t* out) {
  if (naux <= 0 || !L || !out) { set_error("qemb_lpq_upload: bad arguments"); return QEMB_ERR_ARG; }
  DfContext* d = new DfContext();
  int rc = d->set_cholesky_factor(naux, L);
  if (rc) { delete d; return rc; }
  *out = d;
  return QEMB_OK;
}
int qemb_df_create_pbc(int naux, const double* j2c, qemb_df_t* out, int* ischol) {
  if (naux <= 0 || !j2c || !out) { set_error("qemb_df_create_pbc: bad arguments"); return QEMB_ERR_ARG; }
  DfContext* d = new DfContext();
  int rc = d->set_metric_pbc(naux, j2c, ischol);
  if (rc) { delete d; return rc; }
  *out = d;
  return QEMB_OK;
}
int qemb_df_alloc_ints(qemb_df_t df, int N) {
  if (!df) { set_error("qemb_df_alloc_ints: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->alloc_ints(N);
}
int qemb_df_add_pw_block(qemb_df_t df, int nG, const double* F_re, const double* F_im, const double* pw_re, const double* pw_im) {
  if (!df) { set_error("qemb_df_add_pw_block: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->add_pw_block(nG, F_re, F_im, pw_re, pw_im);
}
int qemb_df_add_rs_block(qemb_df_t df, int p0, int p1, const double* block) {
  if (!df) { set_error("qemb_df_add_rs_block: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->add_rs_block(p0, p1, block);
}
int qemb_df_pw_imag_absmax(qemb_df_t df, double* out) {
  if (!df) { set_error("qemb_df_pw_imag_absmax: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->imag_absmax(out);
}
int qemb_df_pw_select(qemb_df_t df, int part) {
  if (!df) { set_error("qemb_df_pw_select: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->select_part(part);
}
int qemb_df_free(qemb_df_t df) { delete reinterpret_cast<DfContext*>(df); return QEMB_OK; }
int qemb_df_set_ints(qemb_df_t df, int N, const double* ints, int layout) {
  if (!df || !ints || N <= 0) { set_error("qemb_df_set_ints: bad arguments"); return QEMB_ERR_ARG; }
  DfContext* d = reinterpret_cast<DfContext*>(df);
  if (layout == 0) return d->set_ints_pqL(N, ints);
  if (layout == 1) return d->set_ints_Lpq(N, ints);
  if (layout == 2) return d->set_ints_packed(N, ints);
  set_error("qemb_df_set_ints: layout must be 0, 1 or 2");
  return QEMB_ERR_ARG;
}
int qemb_df_set_ints_semisparse(qemb_df_t df, int N, int64_t n_unique, const double* unique_dense_data, const int64_t* reach_ptr,
                                const int32_t* reach_nu, const int64_t* reach_off) {
  if (!df) { set_error("qemb_df_set_ints_semisparse: null handle"); return QEMB_ERR_ARG; }
  return reinterpret_cast<DfContext*>(df)->set_ints_semisparse(N, n_unique, unique_dense_data, reach_ptr, reach_nu, reach_off);
}
int qemb_df_transform_screened(qemb_df_t df, const double* TA, int n, const double* S_abs, double MO_coeff_epsilon,
                               double* out_s4_host, qemb_frag_t frag) {
  if (!df || !TA || !S_abs) { set_error("qemb_df_transform_screened: null argument"); return QEMB_ERR_ARG; }
  DfContext* d = reinterpret_cast<DfContext*>(df);
  DBuf dTA, dS, s4;
  int rc;
  if ((rc = dTA.alloc((int64_t)d->N * n)) || (rc = dS.alloc((int64_t)d->N * d->N))) return rc;
  if ((rc = dev_h2d(dTA, TA, sizeof(double) * d->N * n)) || (rc = dev_h2d(dS, S_abs, sizeof(double) * d->N * d->N))) return rc;
  if ((rc = s4.alloc(((int64_t)n * (n + 1) / 2) * ((int64_t)n * (n + 1) / 2)))) return rc;
  DBuf bb;
  if ((rc = d->transform(dTA, n, s4, dS, MO_coeff_epsilon, frag ? &bb : nullptr))) return rc;
  return deliver_s4(s4, n, out_s4_host, frag, &bb, d->naux);
}
// the fitted factor alone (no bb^T bb product): the fragment then lives on it (Fragment::adopt_df_only)
static int df_transform_factor(qemb_df_t df, const double* TA, int n, const double* S_abs, double eps, qemb_frag_t frag) {
  if (!df || !TA || !frag) { set_error("qemb_df_transform_factor: null argument"); return QEMB_ERR_ARG; }
  DfContext* d = reinterpret_cast<DfContext*>(df);
  if (FRAG(frag)->n() != n) { set_error("fragment handle has a different n"); return QEMB_ERR_ARG; }
  DBuf dTA, dS, bb;
  int rc;
  if ((rc = dTA.alloc((int64_t)d->N * n))) return rc;
  if ((rc = dev_h2d(dTA, TA, sizeof(double) * d->N * n))) return rc;
  if (S_abs) {
    if ((rc = dS.alloc((int64_t)d->N * d->N))) return rc;
    if ((rc = dev_h2d(dS, S_abs, sizeof(double) * d->N * d->N))) return rc;
  }
  if ((rc = d->transform(dTA, n, nullptr, S_abs ? dS.p : nullptr, eps, &bb))) return rc;
  return FRAG(frag)->adopt_df_only(std::move(bb), d->naux);
}
int qemb_df_transform_factor(qemb_df_t df, const double* TA, int n, qemb_frag_t frag) { return df_transform_factor(df, TA, n, nullptr, 0.0, frag); }
int qemb_df_transform_screened_factor(qemb_df_t df, const double* TA, int n, const double* S_abs, double MO_coeff_epsilon, qemb_frag_t frag) {
  if (!S_abs) { set_error("qemb_df_transform_screened_factor: null argument"); return QEMB_ERR_ARG; }
  return df_transform_factor(df, TA, n, S_abs, MO_coeff_epsilon, frag);
}
int qemb_df_transform(qemb_df_t df, const double* TA, int n, double* out_s4_host, qemb_frag_t frag) {
  if (!df || !TA) { set_error("qemb_df_transform: null argument"); return QEMB_ERR_ARG; }
  DfContext* d = reinterpret_cast<DfContext*>(df);
  DBuf dTA, s4;
  int rc;
  if ((rc = dTA.alloc((int64_t)d->N * n))) return rc;
  if ((rc = dev_h2d(dTA, TA, sizeof(double) * d->N * n))) return rc;
  if ((rc = s4.alloc(((int64_t)n * (n + 1) / 2) * ((int64_t)n * (n + 1) / 2)))) return rc;
  DBuf bb;
  if ((rc = d->transform(dTA, n, s4, nullptr, 0.0, frag ? &bb : nullptr))) return rc;
  return deliver_s4(s4, n, out_s4_host, frag, &bb, d->naux);
}

// ---------------------------------------------------------------- Schmidt ---------------------------
int qemb_schmidt(const double* lmo, int N, int nmo, int nocc, const int64_t* frag_idx, int n_f, double thr, double* TA, int ld,
                 int* n_b, int* sweeps) {
  if (!lmo || !frag_idx || !TA || !n_b) { set_error("qemb_schmidt: null argument"); return QEMB_ERR_ARG; }
  return schmidt_eigh(lmo, N, nmo, nocc, frag_idx, n_f, thr, TA, ld, n_b, sweeps);
}
int qemb_schmidt_subspace(const double* lmo, int N, int nmo, int nocc, const int64_t* frag_idx, int n_f, double thr, double* TA,
                          int ld, int* n_b, int* sweeps) {
  if (!lmo || !frag_idx || !TA || !n_b) { set_error("qemb_schmidt_subspace: null argument"); return QEMB_ERR_ARG; }
  return schmidt_subspace(lmo, N, nmo, nocc, frag_idx, n_f, thr, TA, ld, n_b, sweeps);
}
int qemb_schmidt_svd(const double* rdm, int N, const int64_t* frag_idx, int n_f, double thr, double* TA, int ld, int* n_b,
                     int* sweeps) {
  if (!rdm || !frag_idx || !TA || !n_b) { set_error("qemb_schmidt_svd: null argument"); return QEMB_ERR_ARG; }
  return schmidt_svd(rdm, N, frag_idx, n_f, thr, TA, ld, n_b, sweeps);
}
int qemb_nsocc_guess(const double* Cproj, int n, int nocc, double* P, int* nsocc, double* mo) {
  if (!Cproj || !nsocc || !mo) { set_error("qemb_nsocc_guess: null argument"); return QEMB_ERR_ARG; }
  return nsocc_guess(Cproj, n, nocc, P, nsocc, mo);
}
int qemb_abs_overlap_prim(int nsh, const int* l, const double* ex, const double* xyz, const int64_t* cart0, int64_t ncart, int nroots,
                          const double* roots, const double* weights, double* out) {
  if (nsh <= 0 || ncart <= 0 || nroots <= 0 || !l || !ex || !xyz || !cart0 || !roots || !weights || !out) { set_error("qemb_abs_overlap_prim: bad arguments"); return QEMB_ERR_ARG; }
  for (int i = 0; i < nsh; ++i) if (l[i] < 0 || l[i] > 4 || ex[i] <= 0.0) { set_error("qemb_abs_overlap_prim: shells need 0 <= l <= 4 and a positive exponent"); return QEMB_ERR_ARG; }
  DBuf dex, dxyz, droots, dw, dout, dl, dc0;      // (int arrays ride in double-sized buffers)
  int rc;
  if ((rc = dex.alloc(nsh)) || (rc = dxyz.alloc(3 * (int64_t)nsh)) || (rc = droots.alloc(nroots)) || (rc = dw.alloc(nroots)) ||
      (rc = dout.alloc(ncart * ncart)) || (rc = dl.alloc((nsh + 1) / 2 + 1)) || (rc = dc0.alloc(nsh))) return rc;
  if ((rc = dev_h2d(dex, ex, sizeof(double) * nsh)) || (rc = dev_h2d(dxyz, xyz, sizeof(double) * 3 * nsh)) ||
      (rc = dev_h2d(droots, roots, sizeof(double) * nroots)) || (rc = dev_h2d(dw, weights, sizeof(double) * nroots)) ||
      (rc = dev_h2d(dl, l, sizeof(int) * nsh)) || (rc = dev_h2d(dc0, cart0, sizeof(int64_t) * nsh))) return rc;
  if ((rc = dev_fill(dout, ncart * ncart, 0.0))) return rc;
  if ((rc = dev_abs_overlap_prim(nsh, reinterpret_cast<const int*>(dl.p), dex, dxyz, reinterpret_cast<const int64_t*>(dc0.p), ncart, nroots, droots, dw, dout))) return rc;
  return dev_d2h(out, dout, sizeof(double) * ncart * ncart);
}
int qemb_matmul(int64_t M, int64_t N, int64_t K, const double* A, int transA, const double* B, int transB, double* C) {
  DBuf dA, dB, dC;
  int rc;
  if ((rc = dA.alloc(M * K)) || (rc = dB.alloc(K * N)) || (rc = dC.alloc(M * N))) return rc;
  if ((rc = dev_h2d(dA, A, sizeof(double) * M * K)) || (rc = dev_h2d(dB, B, sizeof(double) * K * N))) return rc;
  // transA: A is stored K x M;  transB: B is stored N x K
  rc = gemm(M, N, K, 1.0, dA, transA ? M : K, !transA, dB, transB ? K : N, transB != 0, 0.0, dC, N);
  if (rc) return rc;
  return dev_d2h(C, dC, sizeof(double) * M * N);
}

}  // extern "C"
