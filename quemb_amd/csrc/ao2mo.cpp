// ao2mo.cpp -- AO -> fragment (embedding basis) two-electron integral transforms on the device.
//
// Dense (row a3): reference `ao2mo.incore.full(eri_, TA, compact=True)` at molbe/mbe.py:1038 -- four quarter
//   transforms, each one TN GEMM  Out[s',(pqr)] = sum_s TA[s,s'] In[(pqr),s]; the index order cycles so the
//   result needs no separate transposes; output packed to the 4-fold (npair x npair) layout of dataset f{I}.
// DF (rows a4/a5): reference molbe/eri_onthefly.py:108-144 (`low = cholesky(j2c)`, `Lqi = Lqp @ TA`,
//   `Lij = Liq @ TA`, `bb = solve_triangular(low, b)`, `eri = bb.T @ bb`, `restore('4')`) and
//   _cpp/eri_sparse_DF.cpp:560-621 (`sym_P_pq`, `L^-1 sym_P_pq`, `X^T X`, i.e. cublasDtrsm + cublasDsyrk at
//   :667-694).  Here L^-1 is formed once per system (resident, like GPU_MatrixHandle :64-107) so the fit is a
//   plain GEMM, and the final contraction runs directly over packed pairs (i >= j), which IS the 4-fold
//   packed result -- no n^2 x n^2 intermediate and no restore step.  288 GB of HBM hold the whole (P|mu nu)
//   tensor, so the reference's memory-driven aux blocking (eri_onthefly.py:18-42) is not needed.
#include "ao2mo.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace qemb {

static inline int64_t npair(int64_t n) { return n * (n + 1) / 2; }

int AoEri::upload(int N_, const double* host, int sym) {
  N = N_;
  const int64_t np = npair(N), n2 = (int64_t)N * N;
  QTRY(s4.alloc(np * np));
  if (sym == 4) return dev_h2d(s4, host, sizeof(double) * np * np);
  if (sym == 8) {
    DBuf s8;
    const int64_t n8 = np * (np + 1) / 2;
    QTRY(s8.alloc(n8));
    QTRY(dev_h2d(s8, host, sizeof(double) * n8));
    return dev_unpack_s8_to_s4(N, s8, s4);
  }
  if (sym == 1) {
    DBuf s1;
    QTRY(s1.alloc(n2 * n2));
    QTRY(dev_h2d(s1, host, sizeof(double) * n2 * n2));
    return dev_pack_s4(N, s1, s4);
  }
  set_error("AoEri::upload: sym must be 1, 4 or 8");
  return QEMB_ERR_ARG;
}

// Both pair symmetries are kept through the four quarter transforms (each the TN GEMM
// Out[x',(rest)] = sum_x TA[x,x'] In[(rest),x]):
//   [mn][k][l] (unpack kl of the resident s4 rows) -> GEMM l -> [l'][mn][k] -> GEMM k -> [k'][l'][mn]
//   -> keep k' >= l' rows -> [(kl)][mn] -> unpack mn -> [(kl)][m][n] -> GEMM n -> [j'][(kl)][m] -> GEMM m -> [i'][j'][(kl)]
//   -> keep i' >= j' rows = the 4-fold packed (npair(n) x npair(n)) result of dataset f{I}.
// 2 N^2 n (npair(N) + npair(n)) + 2 N n^2 (npair(N) + npair(n)) flop: half of the four full quarter transforms.
int ao2mo_dense(const AoEri& ao, const double* TA, int n, double* out_s4) {
  const int64_t N = ao.N;
  if (N <= 0 || n <= 0 || n > N) { set_error("ao2mo_dense: need 0 < n <= N"); return QEMB_ERR_ARG; }
  const int64_t npN = npair(N), npn = npair(n);
  DBuf W1, W2;
  QTRY(W1.alloc(std::max<int64_t>(npN * N * N, (int64_t)n * n * npN)));
  QTRY(W2.alloc(std::max<int64_t>((int64_t)n * npN * N, npn * N * N)));
  TimerScope lap_AO2MO(TIMER_AO2MO);
  const int tcfg = (n > 192 && n <= 224) ? 13 : -1;   // one 224 x 128 tile instead of two padded 128-row tiles
  QTRY(dev_unpack_tril_rows(npN, N, ao.s4, W1));                                                   // [mn][k][l]
  QTRY(gemm(n, npN * N, N, 1.0, TA, n, false, W1, N, true, 0.0, W2, npN * N, 1, 0, 0, 0, tcfg));                     // [l'][mn][k]
  QTRY(gemm_quarter_lower_rows(n, npN, N, TA, W2, W1));                                            // [k'][l'][mn], rows k' >= l' only
  QTRY(dev_unpack_tril_pair_rows(n, N, W1, W2));                                                   // keep k' >= l' rows, unpack mn: W2 = [(kl)][m][n]
  QTRY(gemm(n, npn * N, N, 1.0, TA, n, false, W2, N, true, 0.0, W1, npn * N, 1, 0, 0, 0, tcfg));                     // [j'][(kl)][m]
  QTRY(gemm_quarter_lower_rows(n, npn, N, TA, W1, W2));                                            // [i'][j'][(kl)], rows i' >= j' only
  QTRY(dev_pack_pair_rows(n, npn, W2, out_s4));
  QTRY(lap_AO2MO.close());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
int DfContext::set_metric(int naux_, const double* j2c) {
  naux = naux_;
  DBuf L;
  QTRY(L.alloc((int64_t)naux * naux));
  QTRY(dev_h2d(L, j2c, sizeof(double) * naux * naux));
  QTRY(dev_cholesky_lower(naux, L));
  QTRY(Linv.alloc((int64_t)naux * naux));
  return dev_tri_inverse_lower(naux, L, Linv);
}
int DfContext::set_cholesky_factor(int naux_, const double* Lh) {
  naux = naux_;
  DBuf L;
  QTRY(L.alloc((int64_t)naux * naux));
  QTRY(dev_h2d(L, Lh, sizeof(double) * naux * naux));
  QTRY(Linv.alloc((int64_t)naux * naux));
  return dev_tri_inverse_lower(naux, L, Linv);
}
int DfContext::set_metric_pbc(int naux_, const double* j2c, int* ischol) {
  naux = naux_;
  DBuf A, w, V;
  const int64_t n2 = (int64_t)naux * naux;
  QTRY(A.alloc(n2));
  QTRY(dev_h2d(A, j2c, sizeof(double) * n2));
  QTRY(Linv.alloc(n2));
  int rc = dev_cholesky_lower(naux, A);
  if (rc == QEMB_OK) { if (ischol) *ischol = 1; return dev_tri_inverse_lower(naux, A, Linv); }
  if (rc != QEMB_ERR_NUMERIC) return rc;
  // not positive definite: (P|Q)^{-1/2} restricted to the eigenvalues above 1e-14   (kbe/eri_onthefly.py:40-45)
  if (ischol) *ischol = 0;
  QTRY(dev_h2d(A, j2c, sizeof(double) * n2));
  QTRY(w.alloc(naux)); QTRY(V.alloc(n2));
  QTRY(dev_jacobi_eigh(naux, A, w, V, nullptr));
  {  // The Jacobi driver diagonalises A + sigma (Gershgorin shift): its eigenvalues carry an absolute error of eps * sigma, too coarse for the
     // reference's ABSOLUTE cut at 1e-14.  Take them again as Rayleigh quotients v^T A v with the unshifted metric (second order in the
     // eigenvector error; rounding ~ eps * |A|, what LAPACK's eigh delivers).
    DBuf AV, D;
    QTRY(AV.alloc(n2)); QTRY(D.alloc(n2));
    QTRY(dev_h2d(A, j2c, sizeof(double) * n2));
    QTRY(gemm(naux, naux, naux, 1.0, A, naux, true, V, naux, false, 0.0, AV, naux));
    QTRY(gemm(naux, naux, naux, 1.0, V, naux, false, AV, naux, false, 0.0, D, naux));
    Copy4Desc c{};
    c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = 1; c.dim[3] = naux;
    c.in = D; c.si[0] = 0; c.si[1] = 0; c.si[2] = 0; c.si[3] = naux + 1;
    c.out = w; c.so[0] = 0; c.so[1] = 0; c.so[2] = 0; c.so[3] = 1; c.alpha = 1.0; c.beta = 0.0;
    QTRY(dev_copy4(c));
  }
  std::vector<double> d(naux);
  QTRY(dev_d2h(d.data(), w, sizeof(double) * naux));
  for (double& x : d) x = (x > 1e-14) ? 1.0 / std::sqrt(std::sqrt(x)) : 0.0;      // columns scaled by d^{-1/4}: Vs Vs^T = V d^{-1/2} V^T
  QTRY(dev_h2d(w, d.data(), sizeof(double) * naux));
  QTRY(dev_mul_bcast_rows(naux, naux, V, w));
  return gemm(naux, naux, naux, 1.0, V, naux, true, V, naux, true, 0.0, Linv, naux);
}
int DfContext::alloc_ints(int N_) {
  if (naux <= 0) { set_error("DfContext: set the metric first"); return QEMB_ERR_ARG; }
  if (N_ <= 0) { set_error("DfContext::alloc_ints: N must be positive"); return QEMB_ERR_ARG; }
  N = N_;
  Usp.release(); n_unique = 0;
  const int64_t sz = (int64_t)naux * N * N;
  QTRY(Lpq.alloc(sz)); QTRY(Lpq_im.alloc(sz));
  Lpq_sum.release(); Lact = nullptr;
  QTRY(dev_fill(Lpq, sz, 0.0));
  return dev_fill(Lpq_im, sz, 0.0);
}
int DfContext::add_rs_block(int p0, int p1, const double* h) {
  if (!Lpq.p || !Lpq_im.p) { set_error("DfContext::add_rs_block: alloc_ints first"); return QEMB_ERR_ARG; }
  if (p0 < 0 || p1 > naux || p0 >= p1 || !h) { set_error("DfContext::add_rs_block: need 0 <= p0 < p1 <= naux"); return QEMB_ERR_ARG; }
  const int64_t n2 = (int64_t)N * N, sz = (int64_t)(p1 - p0) * n2;
  DBuf tmp;
  QTRY(tmp.alloc(sz));
  QTRY(dev_h2d(tmp, h, sizeof(double) * sz));
  return axpby(sz, 1.0, tmp, 1.0, Lpq.p + (int64_t)p0 * n2);
}
int DfContext::add_pw_block(int nG, const double* F_re, const double* F_im, const double* pw_re, const double* pw_im) {
  if (!Lpq.p || !Lpq_im.p) { set_error("DfContext::add_pw_block: alloc_ints first"); return QEMB_ERR_ARG; }
  if (nG <= 0 || !F_re || !F_im || !pw_re || !pw_im) { set_error("DfContext::add_pw_block: bad arguments"); return QEMB_ERR_ARG; }
  const int64_t n2 = (int64_t)N * N;
  DBuf fr, fi, pr, pi;
  QTRY(fr.alloc((int64_t)naux * nG)); QTRY(fi.alloc((int64_t)naux * nG));
  QTRY(pr.alloc((int64_t)nG * n2)); QTRY(pi.alloc((int64_t)nG * n2));
  QTRY(dev_h2d(fr, F_re, sizeof(double) * naux * nG)); QTRY(dev_h2d(fi, F_im, sizeof(double) * naux * nG));
  QTRY(dev_h2d(pr, pw_re, sizeof(double) * nG * n2)); QTRY(dev_h2d(pi, pw_im, sizeof(double) * nG * n2));
  // (F_re + i F_im)(P_re + i P_im): four real products over the plane waves of the block   (kbe/eri_onthefly.py:196-199)
  QTRY(gemm(naux, n2, nG, 1.0, fr, nG, true, pr, n2, false, 1.0, Lpq, n2));
  QTRY(gemm(naux, n2, nG, -1.0, fi, nG, true, pi, n2, false, 1.0, Lpq, n2));
  QTRY(gemm(naux, n2, nG, 1.0, fr, nG, true, pi, n2, false, 1.0, Lpq_im, n2));
  return gemm(naux, n2, nG, 1.0, fi, nG, true, pr, n2, false, 1.0, Lpq_im, n2);
}
int DfContext::imag_absmax(double* out) const {
  if (!Lpq_im.p || !out) { set_error("DfContext::imag_absmax: no plane-wave accumulation on this context"); return QEMB_ERR_ARG; }
  DBuf r;
  QTRY(r.alloc(1));
  QTRY(dev_absmax((int64_t)naux * N * N, Lpq_im, r));
  return dev_d2h(out, r, sizeof(double));
}
int DfContext::select_part(int part) {
  if (part == 0) { Lact = nullptr; return QEMB_OK; }
  if (!Lpq.p || !Lpq_im.p) { set_error("DfContext::select_part: no plane-wave accumulation on this context"); return QEMB_ERR_ARG; }
  if (part == 1) { Lact = Lpq_im.p; return QEMB_OK; }
  if (part == 2) {
    const int64_t sz = (int64_t)naux * N * N;
    QTRY(Lpq_sum.alloc(sz));
    QTRY(dcopy(sz, Lpq, Lpq_sum));
    QTRY(axpby(sz, 1.0, Lpq_im, 1.0, Lpq_sum));
    Lact = Lpq_sum.p;
    return QEMB_OK;
  }
  set_error("DfContext::select_part: part must be 0, 1 or 2");
  return QEMB_ERR_ARG;
}
int DfContext::set_ints_Lpq(int N_, const double* h) {
  N = N_;
  Usp.release(); n_unique = 0; Lact = nullptr; Lpq_im.release(); Lpq_sum.release();
  if (naux <= 0) { set_error("DfContext: set the metric first"); return QEMB_ERR_ARG; }
  QTRY(Lpq.alloc((int64_t)naux * N * N));
  return dev_h2d(Lpq, h, sizeof(double) * naux * N * N);
}
int DfContext::set_ints_pqL(int N_, const double* h) {
  N = N_;
  Usp.release(); n_unique = 0; Lact = nullptr; Lpq_im.release(); Lpq_sum.release();
  if (naux <= 0) { set_error("DfContext: set the metric first"); return QEMB_ERR_ARG; }
  DBuf tmp;
  const int64_t n2 = (int64_t)N * N;
  QTRY(tmp.alloc(n2 * naux));
  QTRY(dev_h2d(tmp, h, sizeof(double) * n2 * naux));
  QTRY(Lpq.alloc(n2 * naux));
  // Lpq[L, (p q)] = pqL[(p q), L]
  return perm4(Lpq, tmp, 1, 1, n2, naux, 0, 1, 3, 2);
}
int DfContext::set_ints_packed(int N_, const double* h) {
  N = N_;
  Usp.release(); n_unique = 0; Lact = nullptr; Lpq_im.release(); Lpq_sum.release();
  if (naux <= 0) { set_error("DfContext: set the metric first"); return QEMB_ERR_ARG; }
  DBuf tmp;
  QTRY(tmp.alloc((int64_t)naux * npair(N)));
  QTRY(dev_h2d(tmp, h, sizeof(double) * naux * npair(N)));
  QTRY(Lpq.alloc((int64_t)naux * N * N));
  return dev_unpack_tril_rows(naux, N, tmp, Lpq);
}

int DfContext::set_ints_semisparse(int N_, int64_t n_unique_, const double* unique_host, const int64_t* ptr, const int32_t* nu,
                                   const int64_t* off) {
  if (naux <= 0) { set_error("DfContext: set the metric first"); return QEMB_ERR_ARG; }
  if (N_ <= 0 || n_unique_ < 0 || !ptr || (n_unique_ > 0 && (!unique_host || !nu || !off))) { set_error("set_ints_semisparse: bad arguments"); return QEMB_ERR_ARG; }
  if (ptr[0] != 0) { set_error("set_ints_semisparse: reach_ptr[0] must be 0"); return QEMB_ERR_ARG; }
  for (int mu = 0; mu < N_; ++mu) {
    if (ptr[mu + 1] < ptr[mu]) { set_error("set_ints_semisparse: reach_ptr must be non-decreasing"); return QEMB_ERR_ARG; }
    for (int64_t e = ptr[mu]; e < ptr[mu + 1]; ++e)
      if (nu[e] < 0 || nu[e] >= N_ || off[e] < 0 || off[e] >= n_unique_) { set_error("set_ints_semisparse: partner or offset out of range"); return QEMB_ERR_ARG; }
  }
  N = N_; n_unique = n_unique_;
  reach_ptr.assign(ptr, ptr + N + 1);
  reach_nu.assign(nu, nu + ptr[N]);
  reach_off.assign(off, off + ptr[N]);
  Lpq.release(); Lact = nullptr; Lpq_im.release(); Lpq_sum.release();
  QTRY(Usp.alloc(std::max<int64_t>(1, n_unique * naux)));
  return n_unique > 0 ? dev_h2d(Usp, unique_host, sizeof(double) * n_unique * naux) : QEMB_OK;
}

// bpT[P(i,j)][L'] (pair rows of naux) -> (ij|kl) = sum_L bb[ij][L] bb[kl][L], bb = bpT Linv^T   (eval_via_cholesky, eri_sparse_DF.cpp:611-621)
int DfContext::finish_from_pair_rows(int n, const double* bpT, double* out_s4, DBuf* keep_bb) const {
  const int64_t np = npair(n);
  DBuf bbT;
  QTRY(bbT.alloc(np * naux));
  QTRY(gemm(np, naux, naux, 1.0, bpT, naux, true, Linv, naux, true, 0.0, bbT, naux));
  if (keep_bb) {                                          // the factor in the layout the fragment keeps: [naux][np]
    QTRY(keep_bb->alloc(naux * np));
    QTRY(perm4(*keep_bb, bbT, 1, 1, np, naux, 0, 1, 3, 2));
  }
  if (!out_s4) return 0;                                   // the factor alone (a fragment that lives on it)
  const int64_t nblk = np >= 2048 ? 8 : 1;
  const int64_t w = ((np + nblk - 1) / nblk + 127) / 128 * 128;
  for (int64_t c0 = 0; c0 < np; c0 += w) {
    const int64_t cw = std::min(w, np - c0);
    QTRY(gemm(np - c0, cw, naux, 1.0, bbT.p + c0 * naux, naux, true, bbT.p + c0 * naux, naux, true, 0.0, out_s4 + c0 * np + c0, np));
  }
  if (nblk > 1) QTRY(dev_mirror_lower(np, out_s4, np));
  return 0;
}

// transform_integral on the semi-sparse tensor (_cpp/eri_sparse_DF.cpp:739-751).  The irregular first contraction
// (contract_with_TA_1st :484-532, an AXPY per (mu, i, nu) in the reference) becomes, for a block of AOs mu, ONE batched GEMM:
//   (P|mu i) = sum_{nu in reach(mu)} TA[nu,i] (P|mu nu)   ==   T1[mu][i][P] = TAg[mu]^T Dg[mu],
// Dg[mu][k][:] = aux vector of the k-th partner of mu, TAg[mu][k][:] = TA row of that partner (zero rows pad short lists), both
// gathered by index.  Every stored aux vector is read twice (once per member of its pair) instead of once per (mu, i).
int DfContext::transform_semisparse(const double* TA, int n, double* out_s4, const double* S_abs, double eps, DBuf* keep_bb) const {
  const int64_t np = npair(n);
  DBuf T1, T2, bpT, maskb, Xb, TAact, idx_dev;
  TimerScope lap_DF(TIMER_DF);
  // get_AO_per_MO (:443-465): (P|mu i) exists only where |S_abs TA|(mu,i) >= eps.  AOs that no embedding orbital reaches drop out
  // of BOTH contractions, so the work and the intermediate follow the fragment's footprint, not the size of the molecule.
  std::vector<int64_t> act;
  if (S_abs) {
    QTRY(Xb.alloc((int64_t)N * n)); QTRY(maskb.alloc((int64_t)N * n));
    QTRY(gemm(N, n, N, 1.0, S_abs, N, true, TA, n, false, 0.0, Xb, n));
    QTRY(dev_threshold_mask((int64_t)N * n, Xb, eps, maskb));
    std::vector<double> mh((size_t)N * n);
    QTRY(dev_d2h(mh.data(), maskb, sizeof(double) * N * n));
    for (int64_t mu = 0; mu < N; ++mu) {
      bool any = false;
      for (int i = 0; i < n && !any; ++i) any = mh[(size_t)mu * n + i] != 0.0;
      if (any) act.push_back(mu);
    }
  } else {
    act.resize(N);
    for (int64_t mu = 0; mu < N; ++mu) act[mu] = mu;
  }
  const int64_t Na = (int64_t)act.size();
  if (Na == 0) {                                          // everything screened away: the transformed integrals vanish
    QTRY(dev_fill(out_s4, np * np, 0.0));
    return lap_DF.close();
  }
  QTRY(T1.alloc(Na * n * naux));
  {
    // blocks of (active) AOs whose gathered operands stay under ~1 GiB
    int64_t budget = (int64_t)1 << 27;
    if (const char* e = std::getenv("QEMB_DF_GATHER_BUDGET")) budget = std::max<int64_t>(1, std::atoll(e));   // doubles; tests force several blocks
    int64_t a0 = 0;
    std::vector<int64_t> idxD, idxT;
    DBuf Dg, TAg;
    while (a0 < Na) {
      int64_t Kc = 0, a1 = a0;
      while (a1 < Na) {
        const int64_t k = reach_ptr[act[a1] + 1] - reach_ptr[act[a1]];
        const int64_t Kn = std::max(Kc, std::max<int64_t>(k, 1));
        if (a1 > a0 && (a1 + 1 - a0) * Kn * (naux + n) > budget) break;
        Kc = Kn; ++a1;
      }
      const int64_t B = a1 - a0;
      idxD.assign(B * Kc, -1); idxT.assign(B * Kc, -1);
      for (int64_t b = 0; b < B; ++b) {
        const int64_t e0 = reach_ptr[act[a0 + b]], cnt = reach_ptr[act[a0 + b] + 1] - e0;
        for (int64_t k = 0; k < cnt; ++k) { idxD[b * Kc + k] = reach_off[e0 + k]; idxT[b * Kc + k] = reach_nu[e0 + k]; }
      }
      QTRY(idx_dev.alloc(2 * B * Kc));
      QTRY(dev_h2d(idx_dev, idxD.data(), sizeof(int64_t) * B * Kc));
      QTRY(dev_h2d(idx_dev.p + B * Kc, idxT.data(), sizeof(int64_t) * B * Kc));
      QTRY(Dg.alloc(B * Kc * naux)); QTRY(TAg.alloc(B * Kc * n));
      QTRY(dev_gather_rows(B * Kc, naux, reinterpret_cast<const int64_t*>(idx_dev.p), Usp, naux, Dg));
      QTRY(dev_gather_rows(B * Kc, n, reinterpret_cast<const int64_t*>(idx_dev.p + B * Kc), TA, n, TAg));
      QTRY(gemm(n, naux, Kc, 1.0, TAg, n, false, Dg, naux, false, 0.0, T1.p + a0 * n * naux, naux, B, Kc * n, Kc * naux, (int64_t)n * naux));
      QTRY(dev_sync());                                  // the host index vectors and the scratch are reused by the next block
      a0 = a1;
    }
  }
  // rows of TA (and of the mask) of the active AOs
  QTRY(idx_dev.alloc(Na));
  QTRY(dev_h2d(idx_dev, act.data(), sizeof(int64_t) * Na));
  QTRY(TAact.alloc(Na * n));
  QTRY(dev_gather_rows(Na, n, reinterpret_cast<const int64_t*>(idx_dev.p), TA, n, TAact));
  if (S_abs) {
    QTRY(dev_gather_rows(Na, n, reinterpret_cast<const int64_t*>(idx_dev.p), maskb, n, Xb));      // Xb: compacted mask [a][i]
    QTRY(dev_scale_rows(Na * n, naux, T1, Xb));
  }
  // contract_with_TA_2nd_to_sym_dense (:560-605): for i <= j, sum_{mu in AO_by_MO[i]} TA[mu,j] (P|mu i) -> T2[j][i][P], batched over i
  QTRY(T2.alloc((int64_t)n * n * naux));
  QTRY(gemm(n, naux, Na, 1.0, TAact, n, false, T1, (int64_t)n * naux, false, 0.0, T2, (int64_t)n * naux, n, 0, naux, naux));
  T1.release();
  QTRY(bpT.alloc(np * naux));
  QTRY(dev_pack_pair_rows(n, naux, T2, bpT));            // rows (j >= i) of T2[j][i][:]
  T2.release();
  QTRY(finish_from_pair_rows(n, bpT, out_s4, keep_bb));
  QTRY(lap_DF.close());
  return 0;
}

int DfContext::transform(const double* TA, int n, double* out_s4, const double* S_abs, double eps, DBuf* keep_bb) const {
  if (n <= 0 || n > N) { set_error("df transform: need 0 < n <= N"); return QEMB_ERR_ARG; }
  if (Linv.p && Usp.p) return transform_semisparse(TA, n, out_s4, S_abs, eps, keep_bb);
  if (!Linv.p || !Lpq.p) { set_error("DfContext: metric and 3-index integrals must be set"); return QEMB_ERR_ARG; }
  const int64_t np = npair(n);
  DBuf T1, T2, bp, bb;
  QTRY(T1.alloc((int64_t)naux * n * N)); QTRY(T2.alloc((int64_t)naux * n * n));
  QTRY(bp.alloc((int64_t)naux * np)); QTRY(bb.alloc((int64_t)naux * np));
  TimerScope lap_DF(TIMER_DF);
  // T1[L,i,nu] = sum_mu TA[mu,i] (L|mu nu)              (eri_onthefly.py:134, batched over L)
  QTRY(gemm(n, N, N, 1.0, TA, n, false, Lact ? Lact : Lpq.p, N, false, 0.0, T1, N, naux, 0, (int64_t)N * N, (int64_t)n * N));
  if (S_abs) {
    // semi-sparse semantics (eri_sparse_DF.cpp:443-532): AO_by_MO[i] = {mu : |S_abs TA|(mu,i) >= eps}; (P|mu i) exists
    // only for mu in AO_by_MO[i].  Pair screening of (P|mu nu) itself arrives as zeros in the packed input.
    DBuf X, mask;
    QTRY(X.alloc((int64_t)n * N)); QTRY(mask.alloc((int64_t)n * N));
    // X[i,mu] = sum_nu TA[nu,i] S_abs[nu,mu]   (S_abs symmetric)
    QTRY(gemm(n, N, N, 1.0, TA, n, false, S_abs, N, false, 0.0, X, N));
    QTRY(dev_threshold_mask((int64_t)n * N, X, eps, mask));
    QTRY(dev_mul_bcast_rows(naux, (int64_t)n * N, T1, mask));
  }
  if (S_abs) {
    // contract_with_TA_2nd_to_sym_dense (eri_sparse_DF.cpp:586-595): for the pair (i <= j) the AO list of the SMALLER index i is
    // walked, sum_{mu in AO_by_MO[i]} TA[mu,j] (P|mu i).  T2[L][j][i] = sum_nu TA[nu,j] T1[L,i,nu], whose lower triangle
    // (row j >= column i) is exactly that element.
    QTRY(gemm(n, n, N, 1.0, TA, n, false, T1, N, true, 0.0, T2, n, naux, 0, (int64_t)n * N, (int64_t)n * n));
  } else {
    // T2[(L,i),j] = sum_nu T1[(L,i),nu] TA[nu,j]           (eri_onthefly.py:136; symmetric in i, j)
    QTRY(gemm((int64_t)naux * n, n, N, 1.0, T1, N, true, TA, n, false, 0.0, T2, n));
  }
  // unique pairs i >= j                                   (eri_sparse_DF.cpp:560-605 sym_P_pq)
  QTRY(dev_pack_tril_rows(naux, n, T2, bp));
  // bb = L^-1 bp                                          (eri_onthefly.py:141 / cublasDtrsm :667)
  QTRY(gemm(naux, np, naux, 1.0, Linv, naux, true, bp, np, false, 0.0, bb, np));
  // (ij|kl) = sum_L bb[L,ij] bb[L,kl] over packed pairs   (eri_onthefly.py:143 / cublasDsyrk :684, beta = 0)
  if (out_s4) QTRY(df_pair_product(np, naux, bb, out_s4));      // (null: the caller wants the factor alone -- a fragment that lives on it)
  if (keep_bb) *keep_bb = std::move(bb);                  // B_{ij}^{L} itself: the fragment's 3-index factor (MO integrals straight from it, ccsd.cpp)
  QTRY(lap_DF.close());
  return 0;
}

// out[P1][P2] = sum_L bb[L,P1] bb[L,P2] (np x np, symmetric) from the packed factor bb[naux][np].  Only block columns at and below
// the diagonal are computed (9/16 of the flops with 8 blocks), then mirrored.
int df_pair_product(int64_t np, int64_t naux, const double* bb, double* out) {
  const int64_t nblk = np >= 2048 ? 8 : 1;
  const int64_t w = ((np + nblk - 1) / nblk + 127) / 128 * 128;
  for (int64_t c0 = 0; c0 < np; c0 += w) {
    const int64_t cw = std::min(w, np - c0);
    QTRY(gemm(np - c0, cw, naux, 1.0, bb + c0, np, false, bb + c0, np, false, 0.0, out + c0 * np + c0, np));
  }
  if (nblk > 1) QTRY(dev_mirror_lower(np, out, np));
  return 0;
}

}  // namespace qemb
