// ao2mo.h -- AO -> embedding-basis ERI transforms (rows a3, a4, a5 of SURVEY.md section 8).
#pragma once
#include <cstdint>
#include <vector>
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
  // Periodic metric (kbe/eri_onthefly.py:19-45 _j2c_cholesky_or_eig): Cholesky when (P|Q) is positive definite, otherwise the fit matrix
  // V_+ d_+^{-1/2} V_+^T over the eigenpairs with d > 1e-14.  Either way `Linv` ends up as the matrix bb = Linv b is formed with.
  int set_metric_pbc(int naux_, const double* j2c_host, int* ischol);
  // Gamma-point CC-GDF accumulation of the fitted 3-index tensor at the AO level (kbe/eri_onthefly.py:160-217):
  //   (L|mu nu) = sum_G F[L,G] (G|mu nu)  +  real-space block rows;   F = ft_ao(chgcell, Gv)^H, (G|mu nu) = ft_aopair * coulG^*
  // The sum over G is complex; its real part lands in Lpq, its imaginary part (zero for a +-G symmetric mesh) in Lpq_im.
  int alloc_ints(int N_);
  int add_rs_block(int p0, int p1, const double* block_host);     // rows [p0, p1) of (L|mu nu) += block (p1 - p0, N, N)
  int add_pw_block(int nG, const double* F_re, const double* F_im, const double* pw_re, const double* pw_im);   // F: naux x nG, pw: nG x N x N
  int imag_absmax(double* out_host) const;
  int select_part(int part);      // which 3-index tensor `transform` reads: 0 real part (default), 1 imaginary part, 2 their sum
  DBuf Lpq_im, Lpq_sum;
  const double* Lact = nullptr;
  int set_ints_pqL(int N_, const double* pqL_host);               // (N, N, naux) as produced by getints3c
  int set_ints_Lpq(int N_, const double* Lpq_host);               // (naux, N, N)
  int set_ints_packed(int N_, const double* P_munu_packed_host);  // (naux, npair(N)), mu >= nu
  // The reference's SemiSparseSym3DTensor (_cpp/eri_sparse_DF.cpp:110-298) as it is, never expanded: `unique` = one aux vector
  // (naux doubles) per stored unique AO pair, n_unique x naux row-major (= the column-major naux x n_unique Eigen matrix);
  // exch_reachable_with_offsets in CSR form (partners nu of mu and the row of the pair's aux vector).  O(n_unique naux) memory.
  int set_ints_semisparse(int N_, int64_t n_unique_, const double* unique_host, const int64_t* reach_ptr, const int32_t* reach_nu,
                          const int64_t* reach_off);
  // S_abs_dev (N x N, may be null) + eps: the MO-coefficient screening of the semi-sparse transform
  // (_cpp/eri_sparse_DF.cpp:443-465 get_AO_per_MO): (P|mu i) is kept only where |S_abs TA|(mu,i) >= eps.
  // keep_bb (nullable): receives the fitted factor B_{ij}^{L} = bb[naux][npair(n)] (eri_onthefly.py:141) the block was formed from
  int transform(const double* TA_dev, int n, double* out_s4_dev, const double* S_abs_dev = nullptr, double eps = 0.0, DBuf* keep_bb = nullptr) const;

  DBuf Usp;                              // semi-sparse storage: [n_unique][naux]
  int64_t n_unique = 0;
  std::vector<int64_t> reach_ptr, reach_off;
  std::vector<int32_t> reach_nu;
 private:
  int transform_semisparse(const double* TA_dev, int n, double* out_s4_dev, const double* S_abs_dev, double eps, DBuf* keep_bb) const;
  int finish_from_pair_rows(int n, const double* bpT, double* out_s4, DBuf* keep_bb) const;
};

// out[np][np] = bb^T bb for the packed factor bb[naux][np] (lower block columns + mirror)
int df_pair_product(int64_t np, int64_t naux, const double* bb, double* out);

}  // namespace qemb
