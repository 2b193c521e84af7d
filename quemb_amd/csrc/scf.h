// scf.h -- device-resident closed-shell RHF for one embedded fragment (rows a6/a7 of SURVEY.md section 8).
#pragma once
#include <cstdint>
#include "dev_ops.h"
#include "tensor_utils.h"

namespace qemb {

struct ScfOptions {
  int max_cycle = 50;            // molbe/helper.py:118
  double conv_tol = 1e-11;       // energy change (PySCF default 1e-9; tighter so the result is a fixed point)
  double conv_tol_grad = 1e-7;   // ||FD - DF||_F
  int diis_space = 8;
  double level_shift = 0.0;
  int verbose = 0;
};

struct ScfResult {
  double e_tot = 0.0;
  bool converged = false;
  int cycles = 0;
};

// J[p,q] = (pq|rs) D[r,s],  K[p,r] = (pq|rs) D[q,s]   (HBM bound).  Without eri_s4: `eri_s1` is the full n^4 tensor.  With
// eri_s4 (the resident npair x npair block): J is contracted from the packed form (a quarter of the bytes) and `eri_s1` must be
// the HALF-unpacked tensor [P(p,q)][r][s] (npair x n x n, half the bytes), from which K is built row pair by row pair.
int build_jk(int n, const double* eri_s1, const double* dm, double* J, double* K, const double* eri_s4 = nullptr);

// The same two matrices from the fragment's 3-index factor B[L][P(p,q)] (eri = B^T B, molbe/eri_onthefly.py:141-143) -- no four-index block is read:
//   J = unpack( B^T (B Dp) ),  Dp the packed density with doubled off-diagonals (two passes over the naux x npair factor);
//   K[p,r] = sum_L sum_{q,s} B_L[p,q] D[q,s] B_L[s,r]: with the occupied orbitals Co (D = 2 Co Co^T), Y_L = Co^T B_L (batched over L) and
//   K = 2 sum_{L,i} Y_L[i,p] Y_L[i,r] -- 2 naux n^2 o + 2 naux o n^2 flops; for a general D, X_L = D^T B_L and K = sum_{L,s} X_L[s,p] B_L[s,r] (2 x 2 naux n^3).
// Bfull: the factor unpacked to [L][p][q] (naux n^2 doubles; the caller keeps it for the cycles of an SCF).  Co (n x o columns of an n x n row-major C) may be
// null: then D is used as it is (any matrix).
int unpack_df_factor(int n, int naux, const double* Bp, double* Bfull);
int build_jk_factor(int n, int naux, const double* Bp, const double* Bfull, const double* dm, const double* C, int o, double* J, double* K);
// where J and K of the fragment RHF come from
struct JkSource {
  const double* eri_s1 = nullptr;   // full or half-unpacked tensor (see build_jk)
  const double* eri_s4 = nullptr;   // resident 4-fold packed block
  const double* Bp = nullptr;       // ... or the 3-index factor (packed pairs), with its unpacked image
  const double* Bfull = nullptr;
  int naux = 0;
};
int build_jk_from(int n, const JkSource& src, const double* dm, const double* C, int o, double* J, double* K);

// h, dm (in: guess, out: converged density), C, eps: device buffers (n*n, n*n, n*n, n).
// J_out/K_out (nullable): J and K of the converged density (n*n each).
int rhf_device(int n, int o, const double* h, const double* eri_s1, double* dm, const ScfOptions& opt, double* C,
               double* eps, double* J_out, double* K_out, ScfResult* res, const double* eri_s4 = nullptr,
               bool c_is_guess = false);
int rhf_device_from(int n, int o, const double* h, const JkSource& src, double* dm, const ScfOptions& opt, double* C,
                    double* eps, double* J_out, double* K_out, ScfResult* res, bool c_is_guess = false);   // c_is_guess: C holds orbitals of a nearby problem (previous sweep): the first
                                           // Fock eigenproblem is rotated into them, like every later cycle into its predecessor

}  // namespace qemb
