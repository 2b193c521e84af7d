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

// h, dm (in: guess, out: converged density), C, eps: device buffers (n*n, n*n, n*n, n).
// J_out/K_out (nullable): J and K of the converged density (n*n each).
int rhf_device(int n, int o, const double* h, const double* eri_s1, double* dm, const ScfOptions& opt, double* C,
               double* eps, double* J_out, double* K_out, ScfResult* res, const double* eri_s4 = nullptr,
               bool c_is_guess = false);   // c_is_guess: C holds orbitals of a nearby problem (previous sweep): the first
                                           // Fock eigenproblem is rotated into them, like every later cycle into its predecessor

}  // namespace qemb
