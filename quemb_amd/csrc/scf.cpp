// scf.cpp -- fragment RHF in the orthonormal embedding basis, entirely on the device.
//
// Reference behaviour: molbe/helper.py:73-151 `get_scfObj` (PySCF `scf.RHF` with custom hcore = fock + heff,
// S = I, `_eri` = fragment ERIs, nelec = 2*nsocc, `dm0`, max_cycle 50, DIIS; level-shift retry :128-149) and
// molbe/helper.py:28-69 `get_veff` (J/K through `scf.hf.dot_eri_dm` :64).  The J/K contractions stream the
// n^4 tensor once each (HBM bound); the Fock eigenproblem goes through the wavefront Jacobi solver.
#include "scf.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace qemb {

static int packed_density(int n, const double* dm, DBuf& Dp);
int build_jk(int n, const double* eri, const double* dm, double* J, double* K, const double* eri_s4) {
  const int64_t n2 = (int64_t)n * n;
  if (eri_s4 && n <= 1024 && (J || K)) {
    // J and K in ONE pass over the 4-fold packed block (4.7 GB at n = 220; the Coulomb product over it plus the exchange build over
    // the 9.4 GB pair-row tensor were two passes, 3.1 ms -> 1.1 ms per build)
    const int64_t np = (int64_t)n * (n + 1) / 2;
    DBuf Dp, Jp;
    if (J) {
      QTRY(Jp.alloc(np));
      QTRY(packed_density(n, dm, Dp));                        // Dp[rs] = D[r,s] + D[s,r] (r > s), D[r,r]
    }
    QTRY(dev_jk_from_packed(n, eri_s4, dm, J ? Dp.p : nullptr, J ? Jp.p : nullptr, K));
    if (J) QTRY(dev_unpack_tril_rows(1, n, Jp, J));
    return 0;
  }
  if (J && eri_s4) {
    // J from the 4-fold packed block (a quarter of the bytes): Jp = eri_s4 . Dp, Dp[rs] = D[r,s] + D[s,r] (r > s), D[r,r]
    const int64_t np = (int64_t)n * (n + 1) / 2;
    DBuf Dp, Jp;
    QTRY(Jp.alloc(np));
    QTRY(packed_density(n, dm, Dp));
    QTRY(dev_gemv_rows(np, np, eri_s4, np, Dp, Jp, 1.0, 0.0));
    QTRY(dev_unpack_tril_rows(1, n, Jp, J));
  } else if (J) {
    QTRY(dev_gemv_rows(n2, n2, eri, n2, dm, J, 1.0, 0.0));
  }
  // K[p,r] = sum_{q,s} D[q,s] (pq|sr).  With eri_s4 given, `eri` is the half-unpacked tensor [P(p,q)][r][s] (half the bytes
  // of the n^4 tensor, and the very operand the MO transformation starts from); otherwise the full s1 tensor [p][(q,s)][r].
  if (K && eri_s4) QTRY(dev_k_from_pairs(n, eri, dm, K));
  else if (K) QTRY(dev_contract_mid(n, n2, n, eri, dm, K, n, 1.0, 0.0));
  return 0;
}

// packed density with doubled off-diagonals: Dp[P(r,s)] = D[r,s] + D[s,r] (r > s), D[r,r]
static int packed_density(int n, const double* dm, DBuf& Dp) {
  QTRY(Dp.alloc((int64_t)n * (n + 1) / 2));
  return dev_pack_density_sym(n, dm, Dp);                 // (one launch; it was four: transpose, sum, diagonal, pack)
}

int unpack_df_factor(int n, int naux, const double* Bp, double* Bfull) { return dev_unpack_tril_rows(naux, n, Bp, Bfull); }

int build_jk_factor(int n, int naux, const double* Bp, const double* Bfull, const double* dm, const double* C, int o, double* J, double* K) {
  const int64_t n2 = (int64_t)n * n, np = (int64_t)n * (n + 1) / 2;
  if (naux <= 0 || !Bp || (K && !Bfull)) { set_error("build_jk_factor: no factor"); return QEMB_ERR_ARG; }
  if (J) {
    DBuf Dp, Jp, z;
    QTRY(packed_density(n, dm, Dp));
    QTRY(Jp.alloc(np)); QTRY(z.alloc(naux));
    QTRY(dev_gemv_rows(naux, np, Bp, np, Dp, z, 1.0, 0.0));                 // z[L] = sum_P B[L,P] Dp[P]
    QTRY(dev_contract_mid(1, naux, np, Bp, z, Jp, np, 1.0, 0.0));            // Jp[P] = sum_L z[L] B[L,P]
    QTRY(dev_unpack_tril_rows(1, n, Jp, J));
  }
  if (K && C && o > 0) {
    // Y[L][i][p] = sum_q C[q,i] B_L[q,p] (batched over L: M = o, N = n, K = n), then K = 2 Y^T Y over the joint index (L,i)
    DBuf Y;
    QTRY(Y.alloc((int64_t)naux * o * n));
    QTRY(gemm(o, n, n, 1.0, C, n, false, Bfull, n, false, 0.0, Y, n, naux, 0, n2, (int64_t)o * n));
    QTRY(gemm(n, n, (int64_t)naux * o, 2.0, Y, n, false, Y, n, false, 0.0, K, n));
  } else if (K) {
    // X[L][s][p] = sum_q D[q,s] B_L[q,p] (batched over L), then K[p,r] = sum_{(L,s)} X[(L,s)][p] B[(L,s)][r]
    DBuf X;
    QTRY(X.alloc((int64_t)naux * n2));
    QTRY(gemm(n, n, n, 1.0, dm, n, false, Bfull, n, false, 0.0, X, n, naux, 0, n2, n2));
    QTRY(gemm(n, n, (int64_t)naux * n, 1.0, X, n, false, Bfull, n, false, 0.0, K, n));
  }
  return 0;
}

int build_jk_from(int n, const JkSource& src, const double* dm, const double* C, int o, double* J, double* K) {
  if (src.Bp && !src.eri_s4 && !src.eri_s1) return build_jk_factor(n, src.naux, src.Bp, src.Bfull, dm, C, o, J, K);
  return build_jk(n, src.eri_s1, dm, J, K, src.eri_s4);
}

static int density_from_mos(int n, int o, const double* C, double* dm) {
  // dm = 2 C_occ C_occ^T : A(m,k) = C[m*n + k] (k < o), B(k,nn) = C[nn*n + k]
  return gemm(n, n, o, 2.0, C, n, true, C, n, true, 0.0, dm, n);
}

static int fock_from_jk(int64_t n2, const double* h, const double* J, const double* K, double* F) {
  const double c[3] = {1.0, 1.0, -0.5};
  const double* xs[3] = {h, J, K};
  return dev_lincomb(n2, 3, c, xs, 0.0, F);
}

static int rhf_loop(int n, int o, const double* h, const JkSource& src, double* dm, const ScfOptions& opt,
                    double* C, double* eps, double* J, double* K, ScfResult* res, bool c_is_guess) {
  const int64_t n2 = (int64_t)n * n;
  DBuf F, Fd, err, tmp, scal, hpf, Cprev, V;
  QTRY(F.alloc(n2)); QTRY(Fd.alloc(n2)); QTRY(err.alloc(n2)); QTRY(tmp.alloc(n2)); QTRY(scal.alloc(4)); QTRY(hpf.alloc(n2));
  QTRY(Cprev.alloc(n2)); QTRY(V.alloc(n2));
  bool have_prev = false;
  if (c_is_guess) { QTRY(dcopy(n2, C, Cprev)); have_prev = true; }
  DeviceDIIS diis(opt.diis_space, n2);
  QTRY(diis.init());
  double e_old = 0.0;
  res->converged = false;
  int cyc = 0;
  bool dm_from_C = false;      // dm = 2 Co Co^T of the orbitals in C (every cycle after the first): the exchange matrix of a factor-resident fragment then needs o columns only
  // Small fragments (n <= dev_scf_fused_max(), round 5): the cycle is launch bound -- ~28 launches and three waits -- so its two ends are single launches each: Fock
  // matrix + energy + commutator + its norm (dev_scf_fock_small), and rotation into the last orbitals + Jacobi + rotation back + copy + density
  // (dev_jacobi_eigh_in_basis, asynchronous: its status word travels with the next cycle's scalars).
  const bool fused = n <= dev_scf_fused_max();
  int* jac_status = reinterpret_cast<int*>(scal.p + 2);
  bool status_pending = false;
  auto check_status = [&](const double* word) {
    int st; std::memcpy(&st, word, sizeof(int));
    if (st < 0) { set_error("Jacobi sweeps did not converge in 40 sweeps"); return (int)QEMB_ERR_NOCONV; }
    return 0;
  };
  static const bool trace = std::getenv("QEMB_SCF_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (cyc = 0; cyc < opt.max_cycle; ++cyc) {
    const double tc0 = trace ? now() : 0.0;
    QTRY(build_jk_from(n, src, dm, dm_from_C ? C : nullptr, o, J, K));
    if (trace) { (void)dev_sync(); std::fprintf(stderr, "[qemb scf trace]   cycle %d: J/K issued + done %.0f us\n", cyc, now() - tc0); }
    if (fused) {
      QTRY(dev_scf_fock_small(n, h, J, K, dm, F, err, scal));
    } else {
    QTRY(fock_from_jk(n2, h, J, K, F));                                   // F = h + J - K/2, one pass (small fragments are launch bound: five launches before)
    QTRY(lincomb2(n2, 1.0, h, 1.0, F, hpf));
    QTRY(dev_dot(n2, hpf, dm, scal));                                   // 2 E = <h + F, D>
    QTRY(gemm_nn(n, n, n, 1.0, F, dm, 0.0, err));                       // FD - DF   (S = I)
    QTRY(gemm_nn(n, n, n, -1.0, dm, F, 1.0, err));
    QTRY(dev_dot(n2, err, err, scal.p + 1));
    }
    double hs[3] = {0.0, 0.0, 0.0};
    QTRY(dev_d2h(hs, scal, sizeof(double) * (status_pending ? 3 : 2)));
    if (status_pending) { status_pending = false; QTRY(check_status(hs + 2)); }
    if (trace) std::fprintf(stderr, "[qemb scf trace]   cycle %d: scalars on the host %.0f us after the cycle began\n", cyc, now() - tc0);
    const double e_tot = 0.5 * hs[0], gnorm = std::sqrt(hs[1]);
    if (opt.verbose > 0) std::fprintf(stderr, "[qemb scf] cycle %2d  E = %.12f  dE = %.3e  |FD-DF| = %.3e\n", cyc, e_tot, e_tot - e_old, gnorm);
    res->e_tot = e_tot;
    if (!std::isfinite(e_tot)) { set_error("fragment SCF diverged"); return QEMB_ERR_NUMERIC; }
    if (cyc > 0 && std::fabs(e_tot - e_old) < opt.conv_tol && gnorm < opt.conv_tol_grad) { res->converged = true; break; }
    e_old = e_tot;
    QTRY(dcopy(n2, F, Fd));
    QTRY(diis.extrapolate(Fd, err));
    if (opt.level_shift != 0.0) {                                        // F += shift * (I - D/2)
      QTRY(axpby(n2, -0.5 * opt.level_shift, dm, 1.0, Fd));
      std::vector<double> ident((size_t)n2, 0.0);
      for (int i = 0; i < n; ++i) ident[(size_t)i * n + i] = opt.level_shift;
      QTRY(dev_h2d(tmp, ident.data(), sizeof(double) * n2));
      QTRY(axpby(n2, 1.0, tmp, 1.0, Fd));
    }
    if (fused) {
      QTRY(dev_jacobi_eigh_in_basis(n, Fd, have_prev ? Cprev.p : nullptr, eps, C, Cprev, o, dm, 1.0e-7, jac_status));
      status_pending = true; have_prev = true; dm_from_C = true;
      if (trace) { (void)dev_sync(); std::fprintf(stderr, "[qemb scf trace]   cycle %d: DIIS + eigensolve done %.0f us after the cycle began\n", cyc, now() - tc0); }
      continue;
    }
    if (have_prev) {
      // rotate into the previous cycle's orbitals first: the Jacobi sweeps start from a nearly diagonal matrix
      QTRY(gemm_nn(n, n, n, 1.0, Fd, Cprev, 0.0, tmp));
      QTRY(gemm_tn(n, n, n, 1.0, Cprev, tmp, 0.0, Fd));
      // (an SCF cycle that is followed by another diagonalisation does not need the last sweep: stopping once the largest rotation of a
      //  sweep is below 1e-7 leaves off-diagonal elements of ~1e-14 relative size, far inside what the next cycle corrects)
      QTRY(dev_jacobi_eigh_until(n, Fd, eps, V, nullptr, 1.0e-7));
      QTRY(gemm_nn(n, n, n, 1.0, Cprev, V, 0.0, C));
    } else {
      QTRY(dev_jacobi_eigh_until(n, Fd, eps, C, nullptr, 1.0e-7));
    }
    QTRY(dcopy(n2, C, Cprev));
    have_prev = true;
    QTRY(density_from_mos(n, o, C, dm));
    dm_from_C = true;
  }
  res->cycles = cyc + (res->converged ? 1 : 0);
  // canonical orbitals of the Fock matrix of the final density (PySCF does the same extra diagonalisation).  On the converged exit F
  // already IS the Fock matrix of dm (the loop left between the convergence test and the density update): no third J/K build.
  if (!res->converged) {
    QTRY(build_jk_from(n, src, dm, dm_from_C ? C : nullptr, o, J, K));
    QTRY(fock_from_jk(n2, h, J, K, F));
  }
  if (fused) {
    QTRY(dev_jacobi_eigh_in_basis(n, F, have_prev ? Cprev.p : nullptr, eps, C, nullptr, o, dm, 1.0e-10, jac_status));
    double word = 0.0;
    QTRY(dev_d2h(&word, scal.p + 2, sizeof(double)));
    return check_status(&word);
  }
  if (have_prev) {
    // in the orbitals of the last cycle the converged Fock matrix is diagonal up to the SCF residual: the Jacobi sweeps need two or
    // three passes instead of the eight of a cold start
    QTRY(gemm_nn(n, n, n, 1.0, F, Cprev, 0.0, tmp));
    QTRY(gemm_tn(n, n, n, 1.0, Cprev, tmp, 0.0, Fd));
    QTRY(dev_jacobi_eigh(n, Fd, eps, V, nullptr));
    QTRY(gemm_nn(n, n, n, 1.0, Cprev, V, 0.0, C));
  } else {
    QTRY(dcopy(n2, F, Fd));
    QTRY(dev_jacobi_eigh(n, Fd, eps, C, nullptr));
  }
  QTRY(density_from_mos(n, o, C, dm));
  return 0;
}

int rhf_device(int n, int o, const double* h, const double* eri, double* dm, const ScfOptions& opt, double* C, double* eps,
               double* J_out, double* K_out, ScfResult* res, const double* eri_s4, bool c_is_guess) {
  JkSource src;
  src.eri_s1 = eri; src.eri_s4 = eri_s4;
  return rhf_device_from(n, o, h, src, dm, opt, C, eps, J_out, K_out, res, c_is_guess);
}
int rhf_device_from(int n, int o, const double* h, const JkSource& src, double* dm, const ScfOptions& opt, double* C, double* eps,
                    double* J_out, double* K_out, ScfResult* res, bool c_is_guess) {
  if (n <= 0 || o <= 0 || o > n) { set_error("rhf_device: bad dimensions"); return QEMB_ERR_ARG; }
  const int64_t n2 = (int64_t)n * n;
  DBuf Jb, Kb;
  double* J = J_out; double* K = K_out;
  if (!J) { QTRY(Jb.alloc(n2)); J = Jb; }
  if (!K) { QTRY(Kb.alloc(n2)); K = Kb; }
  TimerScope lap_SCF(TIMER_SCF);
  DBuf dm_start;
  QTRY(dm_start.alloc(n2)); QTRY(dcopy(n2, dm, dm_start));
  DBuf c_start;
  if (c_is_guess) { QTRY(c_start.alloc(n2)); QTRY(dcopy(n2, C, c_start)); }
  QTRY(rhf_loop(n, o, h, src, dm, opt, C, eps, J, K, res, c_is_guess));
  if (!res->converged) {
    // molbe/helper.py:128-149: retry with level_shift = 0.2 and a 25-vector DIIS space
    ScfOptions o2 = opt;
    o2.level_shift = 0.2; o2.diis_space = 25;
    QTRY(dcopy(n2, dm_start, dm));
    if (c_is_guess) QTRY(dcopy(n2, c_start, C));
    QTRY(rhf_loop(n, o, h, src, dm, o2, C, eps, J, K, res, c_is_guess));
  }
  QTRY(lap_SCF.close());
  return 0;
}

}  // namespace qemb
