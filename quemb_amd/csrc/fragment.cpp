#include <atomic>
// fragment.cpp -- the per-fragment pipeline on the device:
//   fragment RHF (helper.py:73-151) -> embedding->MO integrals (solver.py:900) -> RCCSD (solver.py:907)
//   -> unrelaxed 1-RDM (ccsd_rdm.py:10-20) back-rotated (solver.py:496-505) -> fragment energy (helper.py:220-339)
// with the fragment ERIs resident in HBM between sweeps (the reference re-reads them from HDF5 every call:
// helper.py:182-189, :303-304).
#include "fragment.h"
#include "ao2mo.h"
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unistd.h>

namespace qemb {

static inline int64_t npair(int64_t n) { return n * (n + 1) / 2; }

int Fragment::set_eri_s4_host(const double* s4) {
  const int64_t np = npair(n_);
  clear_df_factor(); s4_transient_.release();
  QTRY(eri_s4_.alloc(np * np));
  return dev_h2d(eri_s4_, s4, sizeof(double) * np * np);
}
int Fragment::set_eri_s4_dev(const double* s4_dev) {
  const int64_t np = npair(n_);
  clear_df_factor();
  QTRY(eri_s4_.alloc(np * np));
  return dev_d2d(eri_s4_, s4_dev, sizeof(double) * np * np);
}
int Fragment::adopt_eri_s4(DBuf&& s4) {
  const int64_t np = npair(n_);
  if (s4.n != np * np || !s4.p) { set_error("adopt_eri_s4: block of the wrong size"); return QEMB_ERR_ARG; }
  clear_df_factor();
  eri_s4_ = std::move(s4);
  return 0;
}
// ---- the fragment's 3-index factor B[naux][npair(n)] (eri = B^T B): set AFTER the ERIs it belongs to (new ERIs drop it)
void Fragment::clear_df_factor() { df_factor_.release(); df_naux_ = 0; }
// A factor that does not belong to the resident ERIs would give a fragment RHF (which reads the block) and amplitude equations (which would read the
// factor) of two different Hamiltonians, silently.  When the factor is set the WHOLE block is probed: for two fixed pseudo-random vectors x the products
// B^T (B x) and eri_s4 x must agree (two gemv-sized passes; round 4 compared the leading 16 x 16 corner only, which a factor that is stale, truncated or
// of a fragment sharing its first pairs passes).  Any failure on the way leaves the fragment WITHOUT a factor.
int Fragment::check_df_factor() {
  if (!eri_s4_.p) return 0;
  const int64_t np = npair(n_);
  constexpr int NV = 2;
  std::vector<double> x((size_t)(np * NV));
  uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)np;
  for (double& e : x) { st = st * 6364136223846793005ull + 1442695040888963407ull; e = (double)((st >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 52) - 1.0; }
  DBuf X, Z, Y1, Y2;
  QTRY(X.alloc(np * NV)); QTRY(Z.alloc((int64_t)df_naux_ * NV)); QTRY(Y1.alloc(np * NV)); QTRY(Y2.alloc(np * NV));
  QTRY(dev_h2d(X, x.data(), sizeof(double) * np * NV));
  QTRY(gemm((int64_t)df_naux_, NV, np, 1.0, df_factor_, np, true, X, NV, false, 0.0, Z, NV));        // Z = B X
  QTRY(gemm(np, NV, (int64_t)df_naux_, 1.0, df_factor_, np, false, Z, NV, false, 0.0, Y1, NV));       // Y1 = B^T Z
  QTRY(gemm(np, NV, np, 1.0, eri_s4_, np, true, X, NV, false, 0.0, Y2, NV));                          // Y2 = eri_s4 X
  std::vector<double> y1((size_t)(np * NV)), y2((size_t)(np * NV));
  QTRY(dev_d2h(y1.data(), Y1, sizeof(double) * np * NV));
  QTRY(dev_d2h(y2.data(), Y2, sizeof(double) * np * NV));
  double worst = 0.0, scale = 1.0;
  for (size_t k = 0; k < y1.size(); ++k) { worst = std::max(worst, std::fabs(y1[k] - y2[k])); scale = std::max(scale, std::fabs(y2[k])); }
  if (!(worst <= 1e-9 * scale)) {
    set_error("set_df_factor: B^T B x differs from the fragment's ERI block applied to x by " + std::to_string(worst) + " (random probe of the whole block): not the factor of these ERIs");
    return QEMB_ERR_ARG;
  }
  return 0;
}
int Fragment::set_df_factor_host(int naux, const double* B) {
  if (naux <= 0 || !B) { set_error("set_df_factor: need naux > 0 and the factor"); return QEMB_ERR_ARG; }
  clear_df_factor();
  int rc = df_factor_.alloc((int64_t)naux * npair(n_));
  if (rc == 0) { df_naux_ = naux; rc = dev_h2d(df_factor_, B, sizeof(double) * naux * npair(n_)); }
  if (rc == 0) rc = check_df_factor();
  if (rc != 0) clear_df_factor();      // a half-set or refused factor is not kept
  return rc;
}
int Fragment::set_df_factor_dev(int naux, const double* B_dev) {
  if (naux <= 0 || !B_dev) { set_error("set_df_factor: need naux > 0 and the factor"); return QEMB_ERR_ARG; }
  clear_df_factor();
  int rc = df_factor_.alloc((int64_t)naux * npair(n_));
  if (rc == 0) { df_naux_ = naux; rc = dev_d2d(df_factor_, B_dev, sizeof(double) * naux * npair(n_)); }
  if (rc == 0) rc = check_df_factor();
  if (rc != 0) clear_df_factor();
  return rc;
}
int Fragment::set_df_only_host(int naux, const double* B) {
  if (naux <= 0 || !B) { set_error("set_df_only: need naux > 0 and the factor"); return QEMB_ERR_ARG; }
  eri_s4_.release(); s4_transient_.release(); clear_df_factor();
  int rc = df_factor_.alloc((int64_t)naux * npair(n_));
  if (rc == 0) { df_naux_ = naux; rc = dev_h2d(df_factor_, B, sizeof(double) * naux * npair(n_)); }
  if (rc != 0) clear_df_factor();
  return rc;
}
int Fragment::set_df_only_dev(int naux, const double* B_dev) {
  if (naux <= 0 || !B_dev) { set_error("set_df_only: need naux > 0 and the factor"); return QEMB_ERR_ARG; }
  eri_s4_.release(); s4_transient_.release(); clear_df_factor();
  int rc = df_factor_.alloc((int64_t)naux * npair(n_));
  if (rc == 0) { df_naux_ = naux; rc = dev_d2d(df_factor_, B_dev, sizeof(double) * naux * npair(n_)); }
  if (rc != 0) clear_df_factor();
  return rc;
}
int Fragment::adopt_df_only(DBuf&& B, int naux) {
  if (naux <= 0 || !B.p || B.n != (int64_t)naux * npair(n_)) { set_error("adopt_df_only: factor of the wrong size"); return QEMB_ERR_ARG; }
  eri_s4_.release(); s4_transient_.release();
  df_factor_ = std::move(B); df_naux_ = naux;
  return 0;
}
int Fragment::materialize_s4(DBuf& out) {
  if (!df_factor_.p) { set_error("Fragment: no factor to form the ERI block from"); return QEMB_ERR_ARG; }
  const int64_t np = npair(n_);
  QTRY(out.alloc(np * np));
  return df_pair_product(np, df_naux_, df_factor_, out);
}
int Fragment::export_eri_s4(double* s4_host) {
  const int64_t np = npair(n_);
  if (eri_s4_.p) return dev_d2h(s4_host, eri_s4_, sizeof(double) * np * np);
  if (!df_factor_.p) { set_error("fragment has no ERIs"); return QEMB_ERR_ARG; }
  DBuf tmp;
  QTRY(materialize_s4(tmp));
  return dev_d2h(s4_host, tmp, sizeof(double) * np * np);
}
int64_t Fragment::resident_bytes() const {
  int64_t words = eri_s4_.n + df_factor_.n + C_.n + eps_.n + dm_.n + J_.n + K_.n + t_prev_.n + z_prev_.n;
  return 8 * words;
}
int Fragment::adopt_df_factor(DBuf&& B, int naux) {
  if (naux <= 0 || !B.p || B.n != (int64_t)naux * npair(n_)) { set_error("adopt_df_factor: factor of the wrong size"); return QEMB_ERR_ARG; }
  // beside a resident block the factor is only read while its route is the cheaper one (or forced): a large auxiliary set with the route left to the cost
  // rule would keep 8 naux npair bytes that nothing reads (~1 GB per n = 220 fragment at naux = 5000) -- not kept
  if (eri_s4_.p && mo_route_ == -1 && !mo_factor_route_pays(n_, naux)) { B.release(); clear_df_factor(); return 0; }
  df_factor_ = std::move(B); df_naux_ = naux;
  return 0;
}
int Fragment::set_mo_route(int route) {
  if (route < -1 || route > 1) { set_error("set_mo_route: -1 (by cost), 0 (four-index transformation) or 1 (3-index factor)"); return QEMB_ERR_ARG; }
  mo_route_ = route;
  return 0;
}
bool Fragment::use_factor_route() const {
  static const bool off = std::getenv("QEMB_MO_FROM_FACTOR") && std::atoi(std::getenv("QEMB_MO_FROM_FACTOR")) == 0;      // A/B measurements
  if (!df_factor_.p || mo_route_ == 0 || off) return false;
  if (!eri_s4_.p) return true;      // a fragment that lives on its factor: forming the block first would cost what the factor's own product costs
  return mo_route_ == 1 || mo_factor_route_pays(n_, df_naux_);
}
// the half-unpacked tensor [P(p,q)][r][s] before the SCF, when something reads it: it is the first operand of the four-index
// transformation (and the exchange operand beyond n = 1024); the factor route needs neither
int Fragment::scf_operand(DBuf& X1, bool* unpacked) {
  *unpacked = false;
  if (mo_route_ == 1 && !df_factor_.p) { set_error("Fragment: the factor route was requested (set_mo_route 1) and no 3-index factor is set"); return QEMB_ERR_ARG; }
  if (use_factor_route() && (n_ <= 1024 || !eri_s4_.p)) return 0;
  if (!eri_s4_.p) QTRY(materialize_s4(s4_transient_));      // factor-only fragment with the four-index route forced (A/B runs): the block lives for this solve
  QTRY(X1.alloc(mo_transform_work(n_)));
  QTRY(dev_unpack_tril_rows_ld(npair(n_), n_, mo_slab_ld(n_), s4_ptr(), X1));      // rows mo_slab_ld(n) apart (ccsd.cpp)
  *unpacked = true;
  return 0;
}
int Fragment::mo_integrals(int o, int nf, DBuf& X1, bool x1_unpacked, MoIntegrals& ints, bool build_Vl, bool build_T34) {
  DBuf X0;
  QTRY(X0.alloc(mo_transform_work(n_)));
  if (!X1.p) { QTRY(X1.alloc(mo_transform_work(n_))); x1_unpacked = false; }
  last_route_factor_ = use_factor_route();
  if (last_route_factor_) return mo_transform_factor(n_, o, nf, df_naux_, df_factor_, X0, X1, C_, ints, build_Vl, build_T34);
  const int rc = mo_transform(n_, o, nf, s4_ptr(), X0, X1, C_, ints, build_Vl, build_T34, x1_unpacked);
  s4_transient_.release();
  return rc;
}
void Fragment::set_energy_data(const double* h1, const double* veff0, const double* veff, double weight, const int* centers, int ncen) {
  const size_t n2 = (size_t)n_ * n_;
  if (h1) h1_.assign(h1, h1 + n2);
  if (veff0) veff0_.assign(veff0, veff0 + n2);
  if (veff) veff_.assign(veff, veff + n2);
  weight_ = weight;
  centers_.assign(centers, centers + ncen);
}

int Fragment::run_scf(int o, const double* h, const double* dm0, const ScfOptions& opt, double* X0, ScfResult* sres, bool warm) {
  const int64_t n2 = (int64_t)n_ * n_;
  o_ = o;
  // The orbitals of this fragment's previous solve (any sweep) serve as the basis in which the Jacobi eigensolver starts: the Fock
  // matrix of the new sweep is nearly diagonal there (heff moves by a matching step), so the first diagonalisation needs two sweeps
  // instead of eight.  Only the eigensolver's starting point changes -- the SCF still starts from dm0 -- so this does not depend on
  // `warm` (which is about reusing amplitudes, i.e. results).
  (void)warm;
  const bool c_guess = have_C_ && C_.p && dm0 && o == c_nocc_;
  if (!c_guess) { have_C_ = false; QTRY(C_.alloc(n2)); }
  QTRY(eps_.alloc(n_)); QTRY(dm_.alloc(n2)); QTRY(J_.alloc(n2)); QTRY(K_.alloc(n2));
  DBuf hd;
  QTRY(hd.alloc(n2));
  QTRY(dev_h2d_async(hd, h, sizeof(double) * n2));        // (consumed on this context's stream)
  if (dm0) {
    QTRY(dev_h2d_async(dm_, dm0, sizeof(double) * n2));
  } else {   // core guess
    DBuf tmp; QTRY(tmp.alloc(n2)); QTRY(dcopy(n2, hd, tmp));
    QTRY(dev_jacobi_eigh(n_, tmp, eps_, C_, nullptr));
    QTRY(gemm(n_, n_, o, 2.0, C_, n_, true, C_, n_, true, 0.0, dm_, n_));
  }
  JkSource src;
  DBuf Bfull;
  if (s4_ptr()) { src.eri_s1 = X0; src.eri_s4 = s4_ptr(); }
  else {        // the fragment lives on its factor: J / K from B (scf.cpp build_jk_factor)
    QTRY(Bfull.alloc((int64_t)df_naux_ * n2));
    QTRY(unpack_df_factor(n_, df_naux_, df_factor_, Bfull));
    src.Bp = df_factor_; src.Bfull = Bfull; src.naux = df_naux_;
  }
  const int rc = rhf_device_from(n_, o, hd, src, dm_, opt, C_, eps_, J_, K_, sres, c_guess);
  have_C_ = (rc == 0 && sres->converged);
  c_nocc_ = o;
  return rc;
}

int Fragment::hf_veff_from_dm(const double* P_host, double* J_host, double* K_host) {
  if (!has_eris()) { set_error("Fragment: ERIs not set"); return QEMB_ERR_ARG; }
  const int64_t n2 = (int64_t)n_ * n_;
  DBuf X0, P, J, K;
  QTRY(P.alloc(n2)); QTRY(J.alloc(n2)); QTRY(K.alloc(n2));
  QTRY(dev_h2d(P, P_host, sizeof(double) * n2));
  if (factor_only()) {
    QTRY(X0.alloc((int64_t)df_naux_ * n2));
    QTRY(unpack_df_factor(n_, df_naux_, df_factor_, X0));
    QTRY(build_jk_factor(n_, df_naux_, df_factor_, X0, P, nullptr, 0, J, K));
  } else {
    if (n_ > 1024) {      // (up to n = 1024 J and K come from the packed block in one pass: no half-unpacked tensor is needed)
      QTRY(X0.alloc(mo_transform_work(n_)));
      QTRY(dev_unpack_tril_rows(npair(n_), n_, eri_s4_, X0));
    }
    QTRY(build_jk(n_, X0, P, J, K, eri_s4_));
  }
  QTRY(dev_d2h(J_host, J, sizeof(double) * n2));
  QTRY(dev_d2h(K_host, K, sizeof(double) * n2));
  return 0;
}

int Fragment::scf_only(int o, const double* h, const double* dm0, const ScfOptions& opt, double* mo_coeff, double* mo_energy,
                       double* J_host, double* K_host, ScfResult* sres) {
  if (!has_eris()) { set_error("Fragment: ERIs not set"); return QEMB_ERR_ARG; }
  if (o <= 0 || o > n_) { set_error("Fragment: need 0 < nsocc <= n"); return QEMB_ERR_ARG; }
  const int64_t n2 = (int64_t)n_ * n_;
  DBuf X0;
  if (n_ > 1024 && eri_s4_.p) {      // (up to n = 1024 J and K come from the packed block in one pass: no half-unpacked tensor is needed)
    QTRY(X0.alloc(mo_transform_work(n_)));
    QTRY(dev_unpack_tril_rows(npair(n_), n_, eri_s4_, X0));     // half-unpacked [P(p,q)][r][s]
  }
  QTRY(run_scf(o, h, dm0, opt, X0, sres));
  if (mo_coeff) QTRY(dev_d2h(mo_coeff, C_, sizeof(double) * n2));
  if (mo_energy) QTRY(dev_d2h(mo_energy, eps_, sizeof(double) * n_));
  if (J_host) QTRY(dev_d2h(J_host, J_, sizeof(double) * n2));
  if (K_host) QTRY(dev_d2h(K_host, K_, sizeof(double) * n2));
  return 0;
}

// Coupled-perturbed HF (shared/external/cphf_utils.py:12-81, used by the HF Jacobian of the QN optimiser,
// shared/external/optqn.py:456-466):  A = 4 (ia|jb) - (ib|ja) - (ij|ab) - diag(e_i - e_a)  (:27-32),
// B0_p = Co^T v_p Cv (:37-41), u_p = A^-1 B0_p (:70), dP_p = -(Co u_p Cv^T + transpose) (:75-81).
// A is the (positive definite) RHF orbital Hessian, so the solve is Cholesky + triangular inverse + two GEMMs.
int Fragment::cphf_response(int o, const double* h, const double* dm0, const ScfOptions& opt, const double* vpots, int npot,
                            double* dPs) {
  if (!has_eris()) { set_error("Fragment: ERIs not set"); return QEMB_ERR_ARG; }
  if (o <= 0 || o >= n_ || npot <= 0) { set_error("cphf_response: bad arguments"); return QEMB_ERR_ARG; }
  const int n = n_, v = n - o;
  const int64_t n2 = (int64_t)n * n, nov = (int64_t)o * v;
  DBuf X1;
  bool x1_unpacked = false;
  QTRY(scf_operand(X1, &x1_unpacked));
  ScfResult sres;
  QTRY(run_scf(o, h, dm0, opt, X1, &sres));
  if (!sres.converged) { set_error("cphf_response: fragment SCF did not converge"); return QEMB_ERR_NOCONV; }
  std::vector<double> C((size_t)n2), eps((size_t)n);
  QTRY(dev_d2h(C.data(), C_, sizeof(double) * n2));
  QTRY(dev_d2h(eps.data(), eps_, sizeof(double) * n));
  MoIntegrals ints;
  QTRY(mo_integrals(o, 0, X1, x1_unpacked, ints, false, false));
  X1.release();
  DBuf A, L, Linv, d;
  QTRY(A.alloc(nov * nov)); QTRY(d.alloc(nov));
  QTRY(dcopy(nov * nov, ints.ovov, A));
  QTRY(axpby(nov * nov, 0.0, A, 4.0, A));                                              // 4 (ia|jb)
  QTRY(perm4(A, ints.ovov, o, v, o, v, 0, 3, 2, 1, -1.0, 1.0));                         // - (ib|ja)
  QTRY(perm4(A, ints.oovv, o, o, v, v, 0, 2, 1, 3, -1.0, 1.0));                         // - (ij|ab)
  std::vector<double> den((size_t)nov);
  for (int i = 0; i < o; ++i) for (int a = 0; a < v; ++a) den[(size_t)i * v + a] = eps[(size_t)o + a] - eps[(size_t)i];
  QTRY(dev_h2d(d, den.data(), sizeof(double) * nov));
  {
    Copy4Desc c{};
    c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = 1; c.dim[3] = nov;
    c.in = d; c.si[3] = 1; c.out = A; c.so[3] = nov + 1; c.alpha = 1.0; c.beta = 1.0;
    QTRY(dev_copy4(c));                                                               // - diag(e_i - e_a)
  }
  ints = MoIntegrals();
  QTRY(dev_cholesky_lower(nov, A));
  QTRY(Linv.alloc(nov * nov));
  QTRY(dev_tri_inverse_lower(nov, A, Linv));
  A.release();
  // right-hand sides B0[(ia), p]
  std::vector<double> B0((size_t)nov * npot), tmp((size_t)o * n);
  for (int p = 0; p < npot; ++p) {
    const double* vp = vpots + (size_t)p * n2;
    for (int i = 0; i < o; ++i) for (int q = 0; q < n; ++q) { double s = 0; for (int r = 0; r < n; ++r) s += C[(size_t)r * n + i] * vp[(size_t)r * n + q]; tmp[(size_t)i * n + q] = s; }
    for (int i = 0; i < o; ++i) for (int a = 0; a < v; ++a) { double s = 0; for (int q = 0; q < n; ++q) s += tmp[(size_t)i * n + q] * C[(size_t)q * n + o + a]; B0[((size_t)i * v + a) * npot + p] = s; }
  }
  DBuf dB, dY, dU;
  QTRY(dB.alloc(nov * npot)); QTRY(dY.alloc(nov * npot)); QTRY(dU.alloc(nov * npot));
  QTRY(dev_h2d(dB, B0.data(), sizeof(double) * nov * npot));
  QTRY(gemm_nn(nov, npot, nov, 1.0, Linv, dB, 0.0, dY));                                // y = L^-1 B0
  QTRY(gemm_tn(nov, npot, nov, 1.0, Linv, dY, 0.0, dU));                                // u = L^-T y
  std::vector<double> U((size_t)nov * npot);
  QTRY(dev_d2h(U.data(), dU, sizeof(double) * nov * npot));
  std::vector<double> X((size_t)n * v);
  for (int p = 0; p < npot; ++p) {
    double* dP = dPs + (size_t)p * n2;
    for (int r = 0; r < n; ++r) for (int a = 0; a < v; ++a) { double s = 0; for (int i = 0; i < o; ++i) s += C[(size_t)r * n + i] * U[((size_t)i * v + a) * npot + p]; X[(size_t)r * v + a] = s; }
    for (int r = 0; r < n; ++r) for (int q = 0; q < n; ++q) { double s = 0; for (int a = 0; a < v; ++a) s += X[(size_t)r * v + a] * C[(size_t)q * n + o + a]; dP[(size_t)r * n + q] = -s; }
    for (int r = 0; r < n; ++r) for (int q = 0; q < r; ++q) { const double t = dP[(size_t)r * n + q] + dP[(size_t)q * n + r]; dP[(size_t)r * n + q] = dP[(size_t)q * n + r] = t; }
    for (int r = 0; r < n; ++r) dP[(size_t)r * n + r] *= 2.0;
  }
  return 0;
}

int Fragment::prepare_ccsd(int o, const double* h, const double* dm0, const FragmentOptions& opt) {
  if (!has_eris()) { set_error("Fragment: ERIs not set"); return QEMB_ERR_ARG; }
  if (o <= 0 || o >= n_) { set_error("Fragment: need 0 < nsocc < n"); return QEMB_ERR_ARG; }
  retire_solver();
  DBuf X1;
  bool x1_unpacked = false;
  QTRY(scf_operand(X1, &x1_unpacked));
  ScfResult sres;
  QTRY(run_scf(o, h, dm0, opt.scf, X1, &sres));
  if (!sres.converged) { set_error("fragment SCF did not converge (also not with level shift 0.2)"); return QEMB_ERR_NOCONV; }
  MoIntegrals ints;
  QTRY(mo_integrals(o, nf_, X1, x1_unpacked, ints, /*build_Vl=*/true, false));   // measurement hook: dense block available for export
  X1.release();
  cc_.reset(new CcsdSolver());
  QTRY(cc_->setup(std::move(ints), eps_));
  return cc_->init_amps();
}
int Fragment::ccsd_reset() { if (!cc_) { set_error("Fragment: prepare_ccsd first"); return QEMB_ERR_ARG; } return cc_->init_amps(); }
int Fragment::ccsd_iterate(int niter, double* e_corr, double* normt) {
  if (!cc_) { set_error("Fragment: prepare_ccsd first"); return QEMB_ERR_ARG; }
  for (int i = 0; i < niter; ++i) QTRY(cc_->iterate(e_corr, normt));
  return 0;
}

int Fragment::solve(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res,
                    double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1_out, double* t2_out) {
  QTRY(solve_begin(o, h, dm0, opt, eeval, res));
  if (!sp_.no_virtuals) {
    bool conv = false;
    QTRY(cc_->kernel(opt.cc, &res->e_corr_mo, &res->n_iter, &conv));
    res->ccsd_converged = conv;
  }
  return solve_end(mo_coeff, mo_energy, rdm1_emb, rdm1_mo, t1_out, t2_out);
}

// Several fragments in one call: the phases around the CCSD iterations (fragment RHF + MO transformation + set-up; amplitudes -> 1-RDM ->
// energies) run per fragment on one host thread and execution context each, as solver.map_fragments does; the CCSD iterations of all
// fragments run in LOCK STEP on the calling thread (ccsd_kernel_lockstep: one grouped launch per operation for all fragments).
// The per-fragment phases of a batched sweep run on PERSISTENT host threads (round 5): worker f serves fragment f of every phase of every sweep.  Starting six
// std::threads three times per sweep cost ~0.1 ms per phase on the sweep's critical path (the last thread starts ~100 us after the first) -- 0.3 ms of an octane BE2
// sweep of 14 ms.  The pool is never destroyed (its threads sleep on a condition variable; a fork()ed child makes its own); a second caller that finds it busy starts
// plain threads as before.
namespace {
class PhasePool {
 public:
  static PhasePool* acquire(int F) {
    static std::mutex guard;
    static PhasePool* pool = nullptr;
    std::lock_guard<std::mutex> lk(guard);
    if (!pool || pool->pid_ != getpid()) pool = new PhasePool();      // (first use, or the child of a fork: the parent's workers do not exist here)
    if (!pool->busy_.try_lock()) return nullptr;
    pool->grow(F);
    return pool;
  }
  void release() { busy_.unlock(); }
  // fn(f) for f in [0, F) on workers 0 .. F-1; returns when every call has returned
  void run(int F, const std::function<void(int)>& fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; active_ = F; pending_ = F; ++gen_;
    }
    cv_work_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  PhasePool() : pid_(getpid()) {}
  void grow(int F) {
    while ((int)nworkers_ < F) {
      const int id = nworkers_++;
      std::thread([this, id] { loop(id); }).detach();
    }
  }
  void loop(int id) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (id < active_) fn = fn_;
      }
      if (!fn) continue;
      (*fn)(id);
      bool last;
      { std::lock_guard<std::mutex> lk(mu_); last = (--pending_ == 0); }
      if (last) cv_done_.notify_one();
    }
  }
  pid_t pid_;
  std::mutex busy_, mu_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)>* fn_ = nullptr;
  int active_ = 0, pending_ = 0, nworkers_ = 0;
  unsigned long long gen_ = 0;
};
}  // namespace

int Fragment::solve_batch(const std::vector<Fragment*>& frs, const std::vector<int>& o, const std::vector<const double*>& h,
                          const std::vector<const double*>& dm0, const FragmentOptions& opt, int eeval, std::vector<FragmentResult>& res,
                          const std::vector<BatchOutputs>& outs, LockstepStats* stats) {
  const int F = (int)frs.size();
  if (F == 0) return 0;
  static const bool trace = std::getenv("QEMB_BATCH_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  res.assign(F, FragmentResult());
  const int have = dev_ctx_count(F + 1);
  if (have < 0) return have;
  // a backend with a single execution context (the scalar mock of tests/hostcheck) runs the per-fragment phases one after the other
  const bool threaded = have >= F + 1 && F > 1;
  std::vector<int> rc(F, 0);
  std::vector<std::string> msg(F);
  static const bool pool_on = !(std::getenv("QEMB_PHASE_POOL") && std::atoi(std::getenv("QEMB_PHASE_POOL")) == 0);      // (0: a std::thread per fragment and phase, for A/B runs)
  PhasePool* pool = (threaded && pool_on) ? PhasePool::acquire(F) : nullptr;
  struct PoolRelease { PhasePool* p; ~PoolRelease() { if (p) p->release(); } } pool_guard{pool};
  auto per_fragment = [&](auto fn) {
    if (threaded) {
      auto body = [&](int f) {
        rc[f] = dev_ctx_bind(f + 1);
        if (rc[f] == 0) rc[f] = fn(f);
        if (rc[f] != 0) msg[f] = last_error();
      };
      if (pool) pool->run(F, body);
      else {
        std::vector<std::thread> th;
        for (int f = 0; f < F; ++f) th.emplace_back([&, f] { body(f); });
        for (auto& t : th) t.join();
      }
    } else {
      for (int f = 0; f < F; ++f) { rc[f] = fn(f); if (rc[f] != 0) msg[f] = last_error(); }
    }
    int worst = 0;
    for (int f = 0; f < F; ++f) if (rc[f] < 0 && worst == 0) { worst = rc[f]; set_error("fragment " + std::to_string(f) + ": " + msg[f]); }
    return worst;
  };
  // Before the iterations: the fragment RHF per fragment (host round trips inside), then MO integrals + CCSD set-up + starting amplitudes.  The latter is a pure
  // launch sequence (~60 launches per small fragment) and CAN be recorded as a tape per fragment and run merged on the home stream like the iterations
  // (QEMB_TAPE_PREPHASE=1).  Measured in round 5 and left off: recording the sequence anew every sweep (stream capture + node queries) and running the merged
  // sequence serially costs more than six streams issuing it side by side -- octane BE2 begin phase 3.1 -> 3.9 ms, BE3 sweep 26.9 -> 27.8 ms.
  static const bool tape_pre = std::getenv("QEMB_TAPE_PREPHASE") && std::atoi(std::getenv("QEMB_TAPE_PREPHASE")) != 0;
  // (one pass of the host threads for both halves of the begin phase since round 5: a fragment goes on to its integrals as soon as ITS RHF is through instead of
  //  waiting for the slowest RHF of the sweep)
  std::vector<double> ms_scf(F, 0.0);
  std::vector<double> ms_cc(F, 0.0), ms_capture(F, 0.0);      // (trace: the slowest fragment's share of the second half of the begin phase)
  std::vector<dev_tape_t> pre(F, nullptr);
  std::vector<char> taped(F, 0);
  struct PreTapes { std::vector<dev_tape_t>& t; ~PreTapes() { for (dev_tape_t x : t) if (x) (void)dev_tape_destroy(x); } } pre_guard{pre};
  const bool small_all = [&] { for (int f = 0; f < F; ++f) { const int64_t oo = o[f], vv = frs[f]->n_ - o[f]; if (oo * oo * vv * vv > ((int64_t)1 << 22)) return false; } return true; }();
  const bool try_tape = tape_pre && threaded && small_all && opt.relax_density == 0;
  QTRY(per_fragment([&](int f) {
    int r = 0;
    {
      const double ts = now();
      r = frs[f]->solve_begin_scf(o[f], h[f], dm0[f], opt, eeval, &res[f]);
      ms_scf[f] = now() - ts;
      if (r != 0) return r;
    }
    if (try_tape && !frs[f]->sp_.no_virtuals && dev_graph_begin(1) == 0) {
      const int rc_cc = frs[f]->solve_begin_cc(true);
      dev_tape_t t = nullptr;
      const int rc_end = dev_tape_end(&t);
      if (rc_cc == 0 && rc_end == 0 && t) { pre[f] = t; taped[f] = 1; }
      else { if (t) (void)dev_tape_destroy(t); frs[f]->retire_solver(); }      // (what failed shows again in the eager pass below)
    }
    const double t0 = now();
    if (!taped[f]) r = frs[f]->solve_begin_cc(false);
    const double t1 = now();
    if (r == 0 && frs[f]->cc_ && F > 1) r = frs[f]->tape_for_lockstep(F);      // recorded side by side (or kept from the last sweep); a lone fragment keeps its executable graph
    const double t2 = now();
    if (r == 0) r = dev_sync();                                               // the lock-step loop reads this fragment's buffers from another stream
    ms_cc[f] = (t1 - t0) + (now() - t2); ms_capture[f] = t2 - t1;
    return r;
  }));
  {
    std::vector<dev_tape_t> run;
    for (int f = 0; f < F; ++f) if (taped[f]) run.push_back(pre[f]);
    if (!run.empty()) {
      QTRY(dev_tape_run(run.data(), (int)run.size()));      // on the calling thread's (home) stream
      QTRY(dev_sync());
      for (int f = 0; f < F; ++f) if (taped[f]) QTRY(frs[f]->cc_->fetch_energy());
    }
  }
  const double t_begin_done = now();
  std::vector<CcsdSolver*> solvers; std::vector<int> idx, ctx; std::vector<CcsdOptions> copt;
  for (int f = 0; f < F; ++f) if (!frs[f]->sp_.no_virtuals) { solvers.push_back(frs[f]->cc_.get()); idx.push_back(f); ctx.push_back(threaded ? f + 1 : 0); copt.push_back(opt.cc); }
  if (!solvers.empty()) {
    std::vector<double> e; std::vector<int> nit; std::vector<char> conv;
    QTRY(ccsd_kernel_lockstep(solvers, copt, ctx, 0, e, nit, conv, stats));
    for (size_t k = 0; k < idx.size(); ++k) { res[idx[k]].e_corr_mo = e[k]; res[idx[k]].n_iter = nit[k]; res[idx[k]].ccsd_converged = conv[k] != 0; }
  }
  const double t_lock_done = now();
  int warn = 0;
  std::vector<int> rc_end(F, 0);
  const int worst = per_fragment([&](int f) {
    rc_end[f] = frs[f]->solve_end(outs[f].mo_coeff, outs[f].mo_energy, outs[f].rdm1_emb, outs[f].rdm1_mo, outs[f].t1, outs[f].t2);
    return rc_end[f];
  });
  if (trace) std::fprintf(stderr, "[qemb batch] %d fragments: begin %.2f ms (RHF <= %.2f, integrals + set-up <= %.2f, recording <= %.2f), lock-step iterations %.2f ms (tapes %.2f, post %.2f), end %.2f ms\n", F,
                          t_begin_done - t_start, *std::max_element(ms_scf.begin(), ms_scf.end()), *std::max_element(ms_cc.begin(), ms_cc.end()), *std::max_element(ms_capture.begin(), ms_capture.end()),
                          t_lock_done - t_begin_done, stats ? stats->ms_tapes : 0.0, stats ? stats->ms_post : 0.0, now() - t_lock_done);
  if (worst) return worst;
  for (int f = 0; f < F; ++f) if (rc_end[f] > 0) { warn = rc_end[f]; set_error("fragment " + std::to_string(f) + ": " + msg[f]); }
  return warn;
}

int Fragment::solve_begin(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res) {
  QTRY(solve_begin_scf(o, h, dm0, opt, eeval, res));
  return solve_begin_cc(false);
}
int Fragment::solve_begin_scf(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res) {
  sp_ = SolvePending();
  sp_.o = o; sp_.opt = opt; sp_.eeval = eeval; sp_.res = res;
  last_route_factor_ = false;
  if (!has_eris()) { set_error("Fragment: ERIs not set"); return QEMB_ERR_ARG; }
  if (o <= 0 || o > n_) { set_error("Fragment: need 0 < nsocc <= n"); return QEMB_ERR_ARG; }
  const int n = n_, v = n - o;
  // nsocc == n: an embedding space without virtual orbitals.  PySCF's CCSD then has empty amplitude arrays and returns E_corr = 0; the
  // sweep body needs the mean-field results only (density = 2 I in any orthonormal basis, no correlation contribution to the energies).
  const int64_t n2 = (int64_t)n * n;
  static const bool trace = std::getenv("QEMB_SCF_TRACE") != nullptr;      // host time of the steps of this phase on stderr (a measuring aid)
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = trace ? now() : 0.0;
  retire_solver();
  const double t1 = trace ? now() : 0.0;
  // ---- fragment RHF on the half-unpacked tensor [P(p,q)][r][s] (kept: it is the first operand of the MO transformation)
  QTRY(scf_operand(sp_.X1, &sp_.x1_unpacked));         // [P(p,q)][r][s] for the four-index route (kept: its first operand); nothing on the factor route
  const double t2 = trace ? now() : 0.0;
  ScfResult sres;
  QTRY(run_scf(o, h, dm0, opt.scf, sp_.X1, &sres, opt.warm_start != 0));
  if (trace) std::fprintf(stderr, "[qemb scf trace] n=%d retire %.0f us, operand %.0f us, run_scf %.0f us (%d cycles)\n", n, t1 - t0, t2 - t1, now() - t2, sres.cycles);
  res->scf_converged = sres.converged; res->scf_cycles = sres.cycles; res->e_scf = sres.e_tot;
  sp_.no_virtuals = (v == 0);
  if (!sres.converged) {
    set_error("fragment SCF did not converge (also not with level shift 0.2)");
    if (opt.strict) return QEMB_ERR_NOCONV;
    sp_.unconverged = true;
  }
  sp_.C.assign((size_t)n2, 0.0); sp_.eps.assign((size_t)n, 0.0);
  QTRY(dev_d2h(sp_.C.data(), C_, sizeof(double) * n2));
  QTRY(dev_d2h(sp_.eps.data(), eps_, sizeof(double) * n));
  return 0;
}
// ---- integrals + CCSD set-up + starting amplitudes: launches only (allocations come from the context's pool once a sweep has run)
int Fragment::solve_begin_cc(bool defer_energy) {
  const int o = sp_.o, v = n_ - o;
  const FragmentOptions& opt = sp_.opt;
  FragmentResult* res = sp_.res;
  if (sp_.no_virtuals) {
    sp_.X1.release();
    res->e_corr_mo = 0.0; res->n_iter = 0; res->ccsd_converged = true; res->lambda_iters = 0;
    return 0;
  }
  dev_alloc_trace_begin();
  MoIntegrals ints;
  QTRY(mo_integrals(o, sp_.eeval ? nf_ : 0, sp_.X1, sp_.x1_unpacked, ints, /*build_Vl=*/false, /*build_T34=*/opt.relax_density != 0));
  sp_.X1.release();
  retire_solver();
  cc_.reset(new CcsdSolver());
  QTRY(cc_->setup(std::move(ints), eps_));
  if (opt.warm_start && t_prev_.p && t_prev_o_ == o) {
    QTRY(cc_->set_amps(t_prev_.p, t_prev_.p + (int64_t)o * v, defer_energy));
  } else {
    QTRY(cc_->init_amps(defer_energy));
  }
  sp_.alloc_hash = dev_alloc_trace_end();
  return 0;
}

// The recorded update of a lock-step sweep.  Recording costs a stream capture and ~50 node queries per fragment and sweep (0.4-0.5 ms of the begin phase of an
// octane sweep, the six host threads queueing on the runtime's capture lock), and a new tape means a new merge plan in dev_tape_run.  The tape only names
// buffers and sizes: when this solve's allocations landed where the last solve's did (the pool hands blocks back in the order they came -- sp_.alloc_hash), the
// last tape IS this solve's tape.  QEMB_TAPE_CACHE=0: record every time; QEMB_TAPE_CACHE_CHECK=1: record anyway and compare with the kept tape (tests).
static std::atomic<long long> g_tape_reused{0}, g_tape_recorded{0};
void tape_cache_counters(long long* reused, long long* recorded, int reset) {
  if (reused) *reused = g_tape_reused.load();
  if (recorded) *recorded = g_tape_recorded.load();
  if (reset) { g_tape_reused = 0; g_tape_recorded = 0; }
}
int Fragment::tape_for_lockstep(int peers) {
  static const bool cache_on = !(std::getenv("QEMB_TAPE_CACHE") && std::atoi(std::getenv("QEMB_TAPE_CACHE")) == 0);
  const bool check = std::getenv("QEMB_TAPE_CACHE_CHECK") && std::atoi(std::getenv("QEMB_TAPE_CACHE_CHECK")) != 0;
  unsigned long long key = sp_.alloc_hash ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(peers + 1)) ^ ((unsigned long long)sp_.o << 40) ^ ((unsigned long long)n_ << 20);
  if (key == 0) key = 1;
  tape_key_ = key;
  if (std::getenv("QEMB_TAPE_CACHE_TRACE")) std::fprintf(stderr, "[qemb tape cache] fragment %p n %d: key %016llx, kept %016llx\n", (void*)this, n_, key, tape_cache_.tape ? tape_cache_.key : 0ull);
  if (cache_on && tape_cache_.tape && tape_cache_.key == key) {
    if (check) {
      QTRY(cc_->prepare_tape(peers));
      if (cc_->tape() && !dev_tape_equal(cc_->tape(), tape_cache_.tape)) { set_error(std::string("the kept tape differs from a fresh recording at the same buffer layout: ") + last_error()); return QEMB_ERR_DEVICE; }
    }
    cc_->adopt_tape(tape_cache_.tape);
    tape_cache_.tape = nullptr;
    g_tape_reused += 1;
    return 0;
  }
  g_tape_recorded += 1;
  return cc_->prepare_tape(peers);
}
void Fragment::retire_solver() {
  if (cc_) {
    dev_tape_t t = cc_->release_tape();
    if (t) {
      if (tape_key_ != 0) { if (tape_cache_.tape) (void)dev_tape_destroy(tape_cache_.tape); tape_cache_.tape = t; tape_cache_.key = tape_key_; }
      else (void)dev_tape_destroy(t);
    }
  }
  tape_key_ = 0;
  cc_.reset();
}

int Fragment::solve_end(double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1_out, double* t2_out) {
  const int o = sp_.o, n = n_, v = n - o, eeval = sp_.eeval;
  const FragmentOptions& opt = sp_.opt;
  FragmentResult* res = sp_.res;
  const bool no_virtuals = sp_.no_virtuals;
  bool unconverged = sp_.unconverged;
  const int64_t n2 = (int64_t)n * n;
  const std::vector<double>& C = sp_.C; const std::vector<double>& eps = sp_.eps;
  std::vector<double> J((size_t)n2), K((size_t)n2);
  if (!no_virtuals && !res->ccsd_converged) {
    set_error("CCSD did not converge in max_cycle iterations");
    if (opt.strict) return QEMB_ERR_NOCONV;
    unconverged = true;
  }
  // ---- amplitudes to the host as requested; unrelaxed 1-RDM (depends on t1 only)
  std::vector<double> t1((size_t)o * v);
  if (!no_virtuals) QTRY(dev_d2h(t1.data(), cc_->t1(), sizeof(double) * o * v));
  if (t1_out && !t1.empty()) std::memcpy(t1_out, t1.data(), sizeof(double) * o * v);
  if (t2_out && !no_virtuals) QTRY(dev_d2h(t2_out, cc_->t2(), sizeof(double) * (int64_t)o * o * v * v));
  // ---- relax_density: Lambda equations, response 1-RDM and the contraction of the response 2-RDM with the fragment ERIs
  std::vector<double> dm1r, Imat;
  if (opt.relax_density && no_virtuals) {
    dm1r.assign((size_t)n2, 0.0);
    for (int i = 0; i < n; ++i) dm1r[(size_t)i * n + i] = 2.0;
    if (eeval) Imat.assign((size_t)n * nf_, 0.0);
  } else if (opt.relax_density) {
    CcLambda lam(*cc_);
    QTRY(lam.setup());
    if (opt.warm_start && z_prev_.p && z_prev_o_ == o) QTRY(lam.set_guess(z_prev_));
    bool lconv = false;
    QTRY(lam.kernel(opt.lam, &res->lambda_iters, &lconv));
    if (!lconv) {
      set_error("CCSD Lambda equations did not converge in max_cycle iterations");
      if (opt.strict) return QEMB_ERR_NOCONV;
      unconverged = true;
    }
    dm1r.assign((size_t)n2, 0.0);
    if (eeval) Imat.assign((size_t)n * nf_, 0.0);
    QTRY(lam.densities(dm1r.data(), eeval ? cc_->integrals().T34.p : nullptr, nf_, eeval ? Imat.data() : nullptr));
    if (opt.keep_amplitudes || opt.warm_start) {
      QTRY(z_prev_.alloc(lam.n_amp()));
      QTRY(dcopy(lam.n_amp(), lam.z(), z_prev_));
      z_prev_o_ = o;
    }
  }
  // rdm1_mo = [[2 I, t1], [t1^T, 0]]  (shared/external/ccsd_rdm.py:10-20)
  if (rdm1_mo && opt.relax_density) {
    std::memcpy(rdm1_mo, dm1r.data(), sizeof(double) * n2);
  } else if (rdm1_mo) {
    std::memset(rdm1_mo, 0, sizeof(double) * n2);
    for (int i = 0; i < o; ++i) rdm1_mo[(size_t)i * n + i] = 2.0;
    for (int i = 0; i < o; ++i) for (int a = 0; a < v; ++a) { rdm1_mo[(size_t)i * n + o + a] = t1[(size_t)i * v + a]; rdm1_mo[(size_t)(o + a) * n + i] = t1[(size_t)i * v + a]; }
  }
  // rdm_emb = C rdm1 C^T / 2 = Co Co^T + (Co t1 Cv^T + Cv t1^T Co^T)/2   (solver.py:496-505)
  std::vector<double> rdm((size_t)n2, 0.0), hfdm((size_t)n2, 0.0), X((size_t)n * v, 0.0);
  for (int p = 0; p < n; ++p) for (int q = 0; q < n; ++q) { double s = 0; for (int i = 0; i < o; ++i) s += C[(size_t)p * n + i] * C[(size_t)q * n + i]; hfdm[(size_t)p * n + q] = s; }
  for (int p = 0; p < n; ++p) for (int a = 0; a < v; ++a) { double s = 0; for (int i = 0; i < o; ++i) s += C[(size_t)p * n + i] * t1[(size_t)i * v + a]; X[(size_t)p * v + a] = s; }
  for (int p = 0; p < n; ++p) for (int q = 0; q < n; ++q) { double s = 0; for (int a = 0; a < v; ++a) s += X[(size_t)p * v + a] * C[(size_t)q * n + o + a]; rdm[(size_t)p * n + q] = s; }
  for (int p = 0; p < n; ++p) for (int q = 0; q <= p; ++q) {
    const double sym = 0.5 * (rdm[(size_t)p * n + q] + rdm[(size_t)q * n + p]);
    rdm[(size_t)p * n + q] = rdm[(size_t)q * n + p] = hfdm[(size_t)p * n + q] + sym;
  }
  if (opt.relax_density) {   // rdm_emb = C dm1 C^T / 2 with the full response density
    std::vector<double> Y((size_t)n2, 0.0);
    for (int p = 0; p < n; ++p) for (int r = 0; r < n; ++r) { const double c = C[(size_t)p * n + r]; for (int q = 0; q < n; ++q) Y[(size_t)p * n + q] += c * dm1r[(size_t)r * n + q]; }
    for (int p = 0; p < n; ++p) for (int q = 0; q < n; ++q) { double t = 0; for (int r = 0; r < n; ++r) t += Y[(size_t)p * n + r] * C[(size_t)q * n + r]; rdm[(size_t)p * n + q] = 0.5 * t; }
  }
  if (rdm1_emb) std::memcpy(rdm1_emb, rdm.data(), sizeof(double) * n2);
  if (mo_coeff) std::memcpy(mo_coeff, C.data(), sizeof(double) * n2);
  if (mo_energy) std::memcpy(mo_energy, eps.data(), sizeof(double) * n);
  // ---- energies
  if (eeval) {
    if (h1_.empty() || veff0_.empty()) { set_error("Fragment: set_energy_data(h1, veff0, ...) before an energy evaluation"); return QEMB_ERR_ARG; }
    std::vector<double> Z1, Z2;
    if (!opt.relax_density && !no_virtuals) QTRY(cc_->energy_intermediates(Z1, Z2));
    if (no_virtuals) { Z1.assign((size_t)o * nf_, 0.0); Z2.clear(); }
    std::vector<double> e1((size_t)nf_, 0.0), e2((size_t)nf_, 0.0), ec((size_t)nf_, 0.0);
    for (int P = 0; P < nf_; ++P) {
      double s1 = 0, sc = 0;
      for (int Q = 0; Q < n; ++Q) {
        const double d = 2.0 * (rdm[(size_t)P * n + Q] - hfdm[(size_t)P * n + Q]);   // helper.py:286
        s1 += h1_[(size_t)P * n + Q] * d; sc += veff0_[(size_t)P * n + Q] * d;
      }
      e1[P] = s1; ec[P] = sc;
      double s2 = 0;
      if (opt.relax_density) {   // e2_P = 1/4 sum_p' C[P,p'] I[p',P]  (cc_lambda.h)
        for (int q = 0; q < n; ++q) s2 += C[(size_t)P * n + q] * Imat[(size_t)q * nf_ + P];
        e2[P] = 0.25 * s2;
        continue;
      }
      for (int i = 0; i < o; ++i) s2 += C[(size_t)P * n + i] * Z1[(size_t)i * nf_ + P];
      for (int a = 0; a < v; ++a) s2 += C[(size_t)P * n + o + a] * Z2[(size_t)a * nf_ + P];
      e2[P] = 0.5 * s2;
    }
    res->e_frag[0] = res->e_frag[1] = res->e_frag[2] = 0.0;
    for (int c : centers_) { res->e_frag[0] += weight_ * e1[c]; res->e_frag[1] += weight_ * e2[c]; res->e_frag[2] += weight_ * ec[c]; }
    // update_ebe_hf (pfrag.py:327-400) with D = Co Co^T: e2_i = sum_j D_ij (2 J_ij - K_ij), J/K of D = J,K(dm)/2
    if (!veff_.empty()) {
      QTRY(dev_d2h(J.data(), J_, sizeof(double) * n2));
      QTRY(dev_d2h(K.data(), K_, sizeof(double) * n2));
      double ehf = 0.0;
      for (int c : centers_) {
        double a = 0;
        for (int Q = 0; Q < n; ++Q) {
          const double D = hfdm[(size_t)c * n + Q];
          a += 2.0 * h1_[(size_t)c * n + Q] * D + veff_[(size_t)c * n + Q] * D + D * (J[(size_t)c * n + Q] - 0.5 * K[(size_t)c * n + Q]);
        }
        ehf += weight_ * a;
      }
      res->ebe_hf = ehf;
    }
  }
  // ---- keep amplitudes for a warm start of the next sweep
  if ((opt.keep_amplitudes || opt.warm_start) && !no_virtuals) {
    const int64_t na = cc_->n_amp();
    QTRY(t_prev_.alloc(na));
    QTRY(dcopy(na, cc_->t1(), t_prev_));
    t_prev_o_ = o;
  }
  // the next sweep may drive this fragment from a host thread bound to ANOTHER execution context (stream): the kept amplitudes,
  // multipliers and orbitals must be complete before this call returns (no inter-stream ordering exists otherwise)
  QTRY(dev_sync());
  retire_solver();
  return unconverged ? QEMB_WARN_NOCONV : 0;
}

}  // namespace qemb
