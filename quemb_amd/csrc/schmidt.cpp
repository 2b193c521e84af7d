// schmidt.cpp -- Schmidt decomposition of the HF 1-RDM (rows a1, a1', a2 of SURVEY.md section 8).
//
// Reference: molbe/pfrag.py:403-494 `schmidt_decomposition` (eigh of the environment block of D = C_occ C_occ^T,
// bath = eigenvectors with thr < |lambda| < 1 - thr, TA = [I_frag (+) Evec_bath]); kbe/solver.py:9-46
// `schmidt_decomp_svd` (SVD of D[env, frag], bath = left vectors with sigma >= thr); molbe/pfrag.py:208-239
// `get_nsocc`.  The O(N_env^3) / O(N_env n_f^2) eigen/SVD work runs as wavefront Jacobi sweeps on the device
// (linalg_f64.hip); the index bookkeeping (which rows are environment, which eigenvalues are bath) is host logic
// exactly as in the reference.
#include <algorithm>
#include <cmath>
#include <vector>
#include "dev_ops.h"
#include "tensor_utils.h"

namespace qemb {

// lmo: N x nmo row-major (host); AO_in_frag: n_f indices.  TA_out: N x (n_f + n_b) row-major written into a buffer
// with leading dimension ld_out >= n_f + n_b (caller gives N x ld_out).  Returns n_b.
int schmidt_eigh(const double* lmo, int N, int nmo, int nocc, const int64_t* frag, int n_f, double thr, double* TA_out,
                 int ld_out, int* n_b_out, int* sweeps_out) {
  if (N <= 0 || nocc <= 0 || nocc > nmo || n_f <= 0 || n_f > N) { set_error("schmidt: bad dimensions"); return QEMB_ERR_ARG; }
  std::vector<char> isfrag((size_t)N, 0);
  for (int k = 0; k < n_f; ++k) {
    if (frag[k] < 0 || frag[k] >= N || isfrag[(size_t)frag[k]]) { set_error("schmidt: bad fragment index list"); return QEMB_ERR_ARG; }
    isfrag[(size_t)frag[k]] = 1;
  }
  std::vector<int> env;
  for (int i = 0; i < N; ++i) if (!isfrag[(size_t)i]) env.push_back(i);
  const int ne = (int)env.size();
  if (ne == 0) {
    // the fragment is the whole system: the environment block is 0 x 0, no bath (numpy's eigh of the empty block at pfrag.py:468 returns
    // empty arrays); TA_lo_eo is the identity in fragment order
    *n_b_out = 0;
    if (sweeps_out) *sweeps_out = 0;
    if (n_f > ld_out) { set_error("schmidt: output buffer too narrow for n_f + n_b columns"); return QEMB_ERR_ARG; }
    for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
    for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[k] * ld_out + k] = 1.0;
    return 0;
  }
  // C_env = rows of the occupied LMO block that belong to the environment; Denv = C_env C_env^T  (pfrag.py:448-465)
  std::vector<double> cenv((size_t)ne * nocc);
  for (int r = 0; r < ne; ++r) for (int k = 0; k < nocc; ++k) cenv[(size_t)r * nocc + k] = lmo[(size_t)env[(size_t)r] * nmo + k];
  DBuf dC, dD, dw, dV;
  QTRY(dC.alloc((int64_t)ne * nocc)); QTRY(dD.alloc((int64_t)ne * ne)); QTRY(dw.alloc(ne)); QTRY(dV.alloc((int64_t)ne * ne));
  QTRY(dev_h2d(dC, cenv.data(), sizeof(double) * ne * nocc));
  TimerScope lap_SCHMIDT(TIMER_SCHMIDT);
  QTRY(gemm_nt(ne, ne, nocc, 1.0, dC, dC, 0.0, dD));
  QTRY(dev_jacobi_eigh(ne, dD, dw, dV, sweeps_out));                       // pfrag.py:468
  QTRY(lap_SCHMIDT.close());
  std::vector<double> w((size_t)ne), V((size_t)ne * ne);
  QTRY(dev_d2h(w.data(), dw, sizeof(double) * ne));
  QTRY(dev_d2h(V.data(), dV, sizeof(double) * ne * ne));
  std::vector<int> bidx;
  for (int i = 0; i < ne; ++i) if (thr < std::fabs(w[(size_t)i]) && std::fabs(w[(size_t)i]) < 1.0 - thr) bidx.push_back(i);   // :484-486
  const int nb = (int)bidx.size();
  *n_b_out = nb;
  if (n_f + nb > ld_out) { set_error("schmidt: output buffer too narrow for n_f + n_b columns"); return QEMB_ERR_ARG; }
  for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
  for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[k] * ld_out + k] = 1.0;                                                 // :490
  for (int r = 0; r < ne; ++r) for (int b = 0; b < nb; ++b) TA_out[(size_t)env[(size_t)r] * ld_out + n_f + b] = V[(size_t)r * ne + bidx[(size_t)b]];   // :491
  return 0;
}

// Same result as schmidt_eigh for an idempotent D = C_occ C_occ^T at O(N_env n_f^2 + N_env nocc n_f) cost instead of the
// O(N_env^3) eigenproblem: D_env^2 = D_env - D_ef D_fe, hence D_env D_ef = D_ef (1 - D_ff): the column space R of D_ef
// (dimension <= n_f) is invariant under the symmetric D_env, the eigenvectors with 0 < lambda < 1 (the bath) all lie in
// R, and everything in R-perp has lambda in {0, 1}.  So: orthonormalise D_ef (Jacobi SVD of an N_env x n_f matrix),
// project D_env on that basis (n_f x n_f), diagonalise the projection, rotate back.
int schmidt_subspace(const double* lmo, int N, int nmo, int nocc, const int64_t* frag, int n_f, double thr, double* TA_out,
                     int ld_out, int* n_b_out, int* sweeps_out) {
  if (N <= 0 || nocc <= 0 || nocc > nmo || n_f <= 0 || n_f > N) { set_error("schmidt: bad dimensions"); return QEMB_ERR_ARG; }
  std::vector<char> isfrag((size_t)N, 0);
  for (int k = 0; k < n_f; ++k) {
    if (frag[k] < 0 || frag[k] >= N || isfrag[(size_t)frag[k]]) { set_error("schmidt: bad fragment index list"); return QEMB_ERR_ARG; }
    isfrag[(size_t)frag[k]] = 1;
  }
  std::vector<int> env;
  for (int i = 0; i < N; ++i) if (!isfrag[(size_t)i]) env.push_back(i);
  const int ne = (int)env.size();
  if (ne == 0) {
    // the fragment is the whole system: the environment block is 0 x 0, no bath (numpy's eigh of the empty block at pfrag.py:468 returns
    // empty arrays); TA_lo_eo is the identity in fragment order
    *n_b_out = 0;
    if (sweeps_out) *sweeps_out = 0;
    if (n_f > ld_out) { set_error("schmidt: output buffer too narrow for n_f + n_b columns"); return QEMB_ERR_ARG; }
    for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
    for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[k] * ld_out + k] = 1.0;
    return 0;
  }
  if (ne < n_f) return schmidt_eigh(lmo, N, nmo, nocc, frag, n_f, thr, TA_out, ld_out, n_b_out, sweeps_out);
  std::vector<double> cenv((size_t)ne * nocc), cf((size_t)n_f * nocc);
  for (int r = 0; r < ne; ++r) for (int k = 0; k < nocc; ++k) cenv[(size_t)r * nocc + k] = lmo[(size_t)env[(size_t)r] * nmo + k];
  for (int r = 0; r < n_f; ++r) for (int k = 0; k < nocc; ++k) cf[(size_t)r * nocc + k] = lmo[(size_t)frag[r] * nmo + k];
  DBuf dCe, dCf, dDef, ds, dU;
  QTRY(dCe.alloc((int64_t)ne * nocc)); QTRY(dCf.alloc((int64_t)n_f * nocc)); QTRY(dDef.alloc((int64_t)ne * n_f));
  QTRY(ds.alloc(n_f)); QTRY(dU.alloc((int64_t)ne * n_f));
  QTRY(dev_h2d(dCe, cenv.data(), sizeof(double) * ne * nocc));
  QTRY(dev_h2d(dCf, cf.data(), sizeof(double) * n_f * nocc));
  TimerScope lap_SCHMIDT(TIMER_SCHMIDT);
  QTRY(gemm_nt(ne, n_f, nocc, 1.0, dCe, dCf, 0.0, dDef));                       // D_ef = C_env C_f^T
  QTRY(dev_jacobi_svd(ne, n_f, dDef, ds, dU, nullptr, sweeps_out));
  std::vector<double> sv((size_t)n_f);
  QTRY(dev_d2h(sv.data(), ds, sizeof(double) * n_f));
  int r = 0;
  while (r < n_f && sv[(size_t)r] > 1.0e-9) ++r;      // sigma^2 = lambda (1 - lambda): a safe superset of thr < lambda < 1 - thr
  *n_b_out = 0;
  for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
  for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[k] * ld_out + k] = 1.0;
  if (r == 0) { QTRY(lap_SCHMIDT.close()); return 0; }
  // A = U_r^T D_env U_r = Z^T Z,  Z = C_env^T U_r   (nocc x r); U is ne x n_f row-major, its first r columns are used
  DBuf dZ, dA, dw, dY, dB;
  QTRY(dZ.alloc((int64_t)nocc * r)); QTRY(dA.alloc((int64_t)r * r)); QTRY(dw.alloc(r)); QTRY(dY.alloc((int64_t)r * r));
  QTRY(gemm(nocc, r, ne, 1.0, dCe, nocc, false, dU, n_f, false, 0.0, dZ, r));      // Z[k,b] = sum_e C_env[e,k] U[e,b]
  QTRY(gemm(r, r, nocc, 1.0, dZ, r, false, dZ, r, false, 0.0, dA, r));             // A = Z^T Z
  QTRY(dev_jacobi_eigh(r, dA, dw, dY, nullptr));
  QTRY(lap_SCHMIDT.close());
  std::vector<double> w((size_t)r), Y((size_t)r * r), U((size_t)ne * n_f);
  QTRY(dev_d2h(w.data(), dw, sizeof(double) * r));
  QTRY(dev_d2h(Y.data(), dY, sizeof(double) * r * r));
  QTRY(dev_d2h(U.data(), dU, sizeof(double) * ne * n_f));
  std::vector<int> bidx;
  for (int i = 0; i < r; ++i) if (thr < std::fabs(w[(size_t)i]) && std::fabs(w[(size_t)i]) < 1.0 - thr) bidx.push_back(i);
  const int nb = (int)bidx.size();
  *n_b_out = nb;
  if (n_f + nb > ld_out) { set_error("schmidt: output buffer too narrow for n_f + n_b columns"); return QEMB_ERR_ARG; }
  for (int e = 0; e < ne; ++e) for (int b = 0; b < nb; ++b) {
    double acc = 0.0;
    for (int c = 0; c < r; ++c) acc += U[(size_t)e * n_f + c] * Y[(size_t)c * r + bidx[(size_t)b]];
    TA_out[(size_t)env[(size_t)e] * ld_out + n_f + b] = acc;
  }
  return 0;
}

// rdm: N x N (host, real).  kbe/solver.py:9-46.
int schmidt_svd(const double* rdm, int N, const int64_t* frag_in, int n_f, double thr, double* TA_out, int ld_out, int* n_b_out,
                int* sweeps_out) {
  if (N <= 0 || n_f <= 0 || n_f > N) { set_error("schmidt_svd: bad dimensions"); return QEMB_ERR_ARG; }
  std::vector<int> frag((size_t)n_f);
  std::vector<char> isfrag((size_t)N, 0);
  for (int k = 0; k < n_f; ++k) {
    int64_t f = frag_in[k] >= 0 ? frag_in[k] : N + frag_in[k];                                                              // :32
    if (f < 0 || f >= N || isfrag[(size_t)f]) { set_error("schmidt_svd: bad fragment index list"); return QEMB_ERR_ARG; }
    frag[(size_t)k] = (int)f; isfrag[(size_t)f] = 1;
  }
  std::vector<int> env;
  for (int i = 0; i < N; ++i) if (!isfrag[(size_t)i]) env.push_back(i);
  const int ne = (int)env.size();
  if (ne < n_f) {
    // a wide block (fewer environment than fragment sites; scipy's svd takes any shape): the left vectors of G are the right vectors of
    // G^T (n_f x ne), which the one-sided Jacobi solver (rows >= columns) takes.  An empty environment has no bath.
    int nb = 0;
    std::vector<double> s((size_t)ne), V((size_t)ne * ne);
    if (ne > 0) {
      std::vector<double> Gt((size_t)n_f * ne);
      for (int k = 0; k < n_f; ++k) for (int r = 0; r < ne; ++r) Gt[(size_t)k * ne + r] = rdm[(size_t)env[(size_t)r] * N + frag[(size_t)k]];
      DBuf dG, ds, dU, dV;
      QTRY(dG.alloc((int64_t)n_f * ne)); QTRY(ds.alloc(ne)); QTRY(dU.alloc((int64_t)n_f * ne)); QTRY(dV.alloc((int64_t)ne * ne));
      QTRY(dev_h2d(dG, Gt.data(), sizeof(double) * n_f * ne));
      QTRY(dev_jacobi_svd(n_f, ne, dG, ds, dU, dV, sweeps_out));
      QTRY(dev_d2h(s.data(), ds, sizeof(double) * ne));
      QTRY(dev_d2h(V.data(), dV, sizeof(double) * ne * ne));
      for (int k = 0; k < ne; ++k) if (s[(size_t)k] >= thr) ++nb;
    } else if (sweeps_out) {
      *sweeps_out = 0;
    }
    *n_b_out = nb;
    if (n_f + nb > ld_out) { set_error("schmidt_svd: output buffer too narrow"); return QEMB_ERR_ARG; }
    for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
    for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[(size_t)k] * ld_out + k] = 1.0;
    for (int r = 0; r < ne; ++r) for (int b = 0; b < nb; ++b) TA_out[(size_t)env[(size_t)r] * ld_out + n_f + b] = V[(size_t)r * ne + b];
    return 0;
  }
  std::vector<double> G((size_t)ne * n_f);
  for (int r = 0; r < ne; ++r) for (int k = 0; k < n_f; ++k) G[(size_t)r * n_f + k] = rdm[(size_t)env[(size_t)r] * N + frag[(size_t)k]];    // :37
  DBuf dG, ds, dU;
  QTRY(dG.alloc((int64_t)ne * n_f)); QTRY(ds.alloc(n_f)); QTRY(dU.alloc((int64_t)ne * n_f));
  QTRY(dev_h2d(dG, G.data(), sizeof(double) * ne * n_f));
  TimerScope lap_SCHMIDT(TIMER_SCHMIDT);
  QTRY(dev_jacobi_svd(ne, n_f, dG, ds, dU, nullptr, sweeps_out));                                                           // :38
  QTRY(lap_SCHMIDT.close());
  std::vector<double> s((size_t)n_f), U((size_t)ne * n_f);
  QTRY(dev_d2h(s.data(), ds, sizeof(double) * n_f));
  QTRY(dev_d2h(U.data(), dU, sizeof(double) * ne * n_f));
  int nb = 0;
  for (int k = 0; k < n_f; ++k) if (s[(size_t)k] >= thr) ++nb;                                                               // :40
  *n_b_out = nb;
  if (n_f + nb > ld_out) { set_error("schmidt_svd: output buffer too narrow"); return QEMB_ERR_ARG; }
  for (int i = 0; i < N; ++i) for (int c = 0; c < ld_out; ++c) TA_out[(size_t)i * ld_out + c] = 0.0;
  for (int k = 0; k < n_f; ++k) TA_out[(size_t)frag[(size_t)k] * ld_out + k] = 1.0;
  for (int r = 0; r < ne; ++r) for (int b = 0; b < nb; ++b) TA_out[(size_t)env[(size_t)r] * ld_out + n_f + b] = U[(size_t)r * n_f + b];
  return 0;
}

// C_ (n x nocc host) -> P_ = C_ C_^T (n x n), nsocc = round(tr P_), mo = eigenvectors of P_ by DEscending
// eigenvalue (= left singular vectors of C_, pfrag.py:228-239).
int nsocc_guess(const double* Cproj, int n, int nocc, double* P_out, int* nsocc, double* mo_out) {
  DBuf dC, dP, dw, dV;
  QTRY(dC.alloc((int64_t)n * nocc)); QTRY(dP.alloc((int64_t)n * n)); QTRY(dw.alloc(n)); QTRY(dV.alloc((int64_t)n * n));
  QTRY(dev_h2d(dC, Cproj, sizeof(double) * n * nocc));
  QTRY(gemm_nt(n, n, nocc, 1.0, dC, dC, 0.0, dP));
  std::vector<double> P((size_t)n * n);
  QTRY(dev_d2h(P.data(), dP, sizeof(double) * n * n));
  double tr = 0.0;
  for (int i = 0; i < n; ++i) tr += P[(size_t)i * n + i];
  *nsocc = (int)std::lround(tr);
  if (P_out) std::copy(P.begin(), P.end(), P_out);
  QTRY(dev_jacobi_eigh(n, dP, dw, dV, nullptr));
  std::vector<double> V((size_t)n * n);
  QTRY(dev_d2h(V.data(), dV, sizeof(double) * n * n));
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) mo_out[(size_t)r * n + c] = V[(size_t)r * n + (n - 1 - c)];
  return 0;
}

}  // namespace qemb
