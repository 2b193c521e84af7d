// ao2mo.h -- AO -> embedding-basis ERI transforms (rows a3, a4, a5 of SURVEY.md section 8).
#pragma once
#include <cstdint>
#include "dev_ops.h"
#include "tensor_utils.h"

namespace qemb {

// AO ERIs resident on the device in the 4-fold packed (npair(N) x npair(N)) form (built once from s8 / s4 / s1 host
// input): a quarter of the N^4 tensor, and exactly the operand the pair-packed transform starts from.
class AoEri {
 public:
  int N = 0;
  DBuf s4;
  int upload(int N_, const double* eri_host, int sym);   // sym = 8, 4 or 1
};

// (ij|kl) = sum TA_mu,i TA_nu,j TA_ka,k TA_la,l (mu nu|ka la), 4-fold packed output (device, npair(n)^2)
int ao2mo_dense(const AoEri& ao, const double* TA_dev, int n, double* out_s4_dev);

// Density-fitted transform with a resident metric factor.
class DfContext {
 public:
  int N = 0, naux = 0;
  DBuf Linv;    // inverse of the lower Cholesky factor of (P|Q)
  DBuf Lpq;     // (P|mu nu) as [naux][N][N]
  int set_metric(int naux_, const double* j2c_host);              // Cholesky on the device, then invert
  int set_cholesky_factor(int naux_, const double* L_host);       // caller supplies L (lower), just invert
  int set_ints_pqL(int N_, const double* pqL_host);               // (N, N, naux) as produced by getints3c
  int set_ints_Lpq(int N_, const double* Lpq_host);               // (naux, N, N)
  int set_ints_packed(int N_, const double* P_munu_packed_host);  // (naux, npair(N)), mu >= nu
  // S_abs_dev (N x N, may be null) + eps: the MO-coefficient screening of the semi-sparse transform
  // (_cpp/eri_sparse_DF.cpp:443-465 get_AO_per_MO): (P|mu i) is kept only where |S_abs TA|(mu,i) >= eps.
  int transform(const double* TA_dev, int n, double* out_s4_dev, const double* S_abs_dev = nullptr, double eps = 0.0) const;
};

}  // namespace qemb
