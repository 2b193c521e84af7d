// ccsd.cpp -- RCCSD amplitude equations as GEMM-shaped device contractions.
//
// Reference behaviour: molbe/solver.py:829-946 (`solve_ccsd`): PySCF `cc.CCSD(mf)`, `eris = mycc.ao2mo()`
// (:900), `eris.fock = diag(mo_energy)` (:901-902), `mycc.kernel(eris)` (:907).  The equations are PySCF's
// cc/rccsd.py `update_amps` + cc/rintermediates.py as restated in SURVEY.md Appendix A and in
// oracle/qemb_oracle/ccsd.py; here every O(N^5)/O(N^6) term is factorised into dev_gemm calls (FP64 MFMA)
// and the index shuffles between them into dev_copy4 passes.  With the forced-diagonal Fock f_ov = 0 and
// f_oo / f_vv cancel against the orbital-energy shift, so those terms are dropped analytically.
//
// Layout conventions (all contiguous, row-major):
//   t1[i,a]; t2[i,j,a,b]; tau = t2 + t1 (x) t1
//   T [k,c,j,b] = t2[k,j,c,b]   ("ph layout", matrix (kc) x (jb))      Tp[k,c,j,b] = t2[k,j,b,c]
//   W1[(ia),(kc)] = Wvoov[a,k,i,c]     W2[(ia),(kc)] = Wvovo[a,k,c,i]
//   Vl[a,b,c,d] = (ac|bd): the pp-ladder is the NT GEMM  t2new[(ij),(ab)] += tau[(ij),(cd)] * Vl[(ab),(cd)]
//   U accumulates every term that enters t2new as P(X) = X_ijab + X_jiba; it is symmetrised once.
#include "ccsd.h"
#include "ao2mo.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace qemb {

// ------------------------------------------------------------------------------------------------------------
// embedding -> MO transformation on pair-packed operands.  Every step is the TN GEMM
//   Out[x', (rest)] = sum_x C[x,x'] In[(rest), x]      (M = n, N = rest, K = n)
// and both pair symmetries of (pq|rs) are used, so each GEMM runs over npair*n columns instead of n^3
// (half the flops of the full four-index transformation):
//   X1 [pq][r][s]        <- unpack rs of the resident s4 block (p >= q rows only)
//   X0 [s'][pq][r]       <- GEMM over s          X1 [r'][s'][pq]   <- GEMM over r   (symmetric in r',s')
//   X0 [(r's')][p][q]    <- keep r' >= s' rows and unpack pq (one fused pass)
//   X1 [(r's')][P][q']   <- batched GEMM slab.C: the 3/4-transformed integrals (P q'|r' s') that the
//                           fragment-projected energy of get_frag_energy (helper.py:307-321) needs
//   X0 [(r's')][p'][q']  <- batched GEMM C^T.slab: the pair-first MO tensor every block below is gathered from in contiguous runs
// ------------------------------------------------------------------------------------------------------------
// Row stride of the n x n images the two unpack passes write and the K = n products read: the next multiple of 16 doubles, so that the
// 256-byte runs of the tiled unpack sit on whole 128-byte lines (partially written lines cost a read-modify-write at the memory side:
// n = 220 unpacks at 3.7 TB/s with stride 220, at 5.7 TB/s with stride 224).  Small fragments keep the dense layout.
int mo_slab_ld(int n) { return (n >= 64 && n <= 1024) ? (n + 15) / 16 * 16 : n; }
int64_t mo_transform_work(int n) { return (int64_t)n * mo_slab_ld(n) * ((int64_t)n * (n + 1) / 2); }

// The 3/4-transformed integrals the energies need, from T[P(r's')][P][q'] (P = embedding index, rows of n, `slab` doubles per pair)
static int mo_three_quarter_blocks(int n, int o, int nf, const double* T, int64_t slab, MoIntegrals& out, bool build_T34) {
  const int v = n - o;
  if (nf > 0 && build_T34) {   // every (P q'|r' s') as T34[q'][r'][s'][P]: the operand of CcLambda::densities
    QTRY(out.T34.alloc((int64_t)n * n * n * nf));
    QTRY(dev_extract_pf_t(n, T, 0, 0, 0, 0, n, n, n, nf, out.T34, slab));
  } else if (nf > 0) {
    QTRY(out.A1.alloc((int64_t)v * o * v * nf));
    QTRY(out.A2.alloc((int64_t)o * o * v * nf));
    QTRY(dev_extract_pf_t(n, T, o, 0, o, 0, v, o, v, nf, out.A1, slab));     // A1[a,j,b,P] = (P a|j b) = T[P(j,b)][P][a]
    QTRY(dev_extract_pf_t(n, T, 0, 0, o, 0, o, o, v, nf, out.A2, slab));     // A2[i,j,b,P] = (P i|j b)
  }
  return 0;
}

// every block of the amplitude equations from the pair-first MO tensor Mp[P(r',s')][p'][q'] = (r's'|p'q') (slabs of n x n); `scratch` holds v^4 doubles or is null
static int mo_blocks_from_pair_first(int n, int o, const double* Mp, double* scratch, int64_t scratch_elems, MoIntegrals& out, bool build_Vl) {
  const int v = n - o;
  QTRY(out.oooo.alloc((int64_t)o * o * o * o));
  QTRY(out.ovoo.alloc((int64_t)o * v * o * o));
  QTRY(out.ovov.alloc((int64_t)o * v * o * v));
  QTRY(out.oovv.alloc((int64_t)o * o * v * v));
  QTRY(out.ovvo.alloc((int64_t)o * v * v * o));
  QTRY(out.ovvv.alloc((int64_t)o * v * v * v));
  QTRY(dev_extract_pf(n, Mp, 0, 0, 0, 0, o, o, o, o, out.oooo));
  QTRY(dev_extract_pf(n, Mp, 0, o, 0, 0, o, v, o, o, out.ovoo));
  QTRY(dev_extract_pf(n, Mp, 0, o, 0, o, o, v, o, v, out.ovov));
  QTRY(dev_extract_pf(n, Mp, 0, 0, o, o, o, o, v, v, out.oovv));
  QTRY(dev_extract_pf(n, Mp, 0, o, o, 0, o, v, v, o, out.ovvo));
  QTRY(dev_extract_pf(n, Mp, 0, o, o, o, o, v, v, v, out.ovvv));
  {  // (+/-) pair-packed ladder operands (6.4 GB instead of the 12.8 GB dense v^4 block at v = 200)
    const int64_t npv = (int64_t)v * (v + 1) / 2, nm = (int64_t)v * (v - 1) / 2;
    out.ldp = npv + (npv & 1); out.ldm = nm + (nm & 1);
    if (out.ldm == 0) out.ldm = 2;
    QTRY(out.Vp.alloc(npv * out.ldp));
    QTRY(out.Vm.alloc(std::max<int64_t>(nm, 1) * out.ldm));
    QTRY(dev_ladder_pack_vvvv_pf(n, o, Mp, out.Vp, out.ldp, out.Vm, out.ldm));
  }
  if (build_Vl) {  // Vl[a,b,c,d] = (ac|bd): gather [a][c][b][d], then swap the middle indices
    const int64_t v4 = (int64_t)v * v * v * v;
    QTRY(out.Vl.alloc(v4));
    DBuf tmp;
    double* g = scratch;
    if (!g || v4 > scratch_elems) { QTRY(tmp.alloc(v4)); g = tmp; }
    QTRY(dev_extract_pf(n, Mp, o, o, o, o, v, v, v, v, g));
    QTRY(perm4(out.Vl, g, v, v, v, v, 0, 2, 1, 3));
  }
  return 0;
}

int mo_transform(int n, int o, int nf, const double* eri_s4, double* X0, double* X1, const double* C, MoIntegrals& out, bool build_Vl, bool build_T34,
                 bool x1_is_unpacked) {
  const int v = n - o;
  const int64_t np = (int64_t)n * (n + 1) / 2, ncol = np * n;
  out.n = n; out.o = o; out.v = v; out.nf = nf;
  TimerScope lap_AO2MO(TIMER_AO2MO);
  const int64_t nl = mo_slab_ld(n);                   // row stride of the unpacked n x n images (X1 here, X0 after the second unpack)
  if (!x1_is_unpacked) QTRY(dev_unpack_tril_rows_ld(np, n, nl, eri_s4, X1));   // (the caller of solve_begin has done it already)
  // 193..224 rows fit ONE 224 x 128 tile (1.8 % padding instead of the 14 % of two 128-row tiles at n = 220)
  const int tcfg = (n > 192 && n <= 224) ? 13 : -1;
  QTRY(gemm(n, ncol, n, 1.0, C, n, false, X1, nl, true, 0.0, X0, ncol, 1, 0, 0, 0, tcfg));
  QTRY(gemm_quarter_lower_rows(n, np, n, C, X0, X1));     // X1[r'][s'][pq], only the rows r' >= s' (all that is read below)
  QTRY(dev_unpack_tril_pair_rows_ld(n, n, nl, X1, X0));   // keep r' >= s' rows AND unpack pq, one pass: X0 = [(r's')][p][q], rows nl apart
  // the last two quarter transforms act on the n x n slab of every pair (r's'): two batched GEMMs, slab <- C^T slab C, which
  // leave the pair index IN FRONT -- every gather below then reads contiguous runs
  const int64_t n2 = (int64_t)n * n;
  // X1[(r's')][P][q'] = sum_q X0[..][P][q] C[q,q']: the slabs are contiguous, so this is ONE tall product over the np * n rows ((r's'),P) with
  // all n columns in a 128 x 224 tile (as a batch of n x n x n products on the 224 x 128 tile the second column tile is 72 % padding at n = 220)
  if (n > 192 && n <= 224) { QTRY(gemm(np * n, n, n, 1.0, X0, nl, true, C, n, false, 0.0, X1, n, 1, 0, 0, 0, 34)); }
  else QTRY(gemm(n, n, n, 1.0, X0, nl, true, C, n, false, 0.0, X1, n, np, (int64_t)n * nl, 0, n2, tcfg));
  QTRY(mo_three_quarter_blocks(n, o, nf, X1, 0, out, build_T34));
  // X0[(r's')][p'][q'] = sum_p C[p,p'] X1[..][p][q'].  As a batch of n x n x n products on the 224 x 128 tile the second column tile is 72 % padding at n = 220
  // (0.66 of the matrix peak, against 0.74-0.77 for the other two K = n products).  The result is symmetric in (p',q'), so its TRANSPOSE slab by slab is as good:
  //   X0[(r's')][q'][p'] = sum_p X1[(r's')][p][q'] C[p,p']
  // and that is ONE tall product over the rows ((r's'), q') -- the rows of a stack of slabs read through the slab-aware loader (GemmDesc::a_slab) -- with all n
  // columns in a 128 x 224 tile, like the third quarter transform above (round 5).  Other sizes keep the batched form.
  if (n > 192 && n <= 224 && n % 2 == 0) {
    GemmDesc g{};
    g.M = np * n; g.N = n; g.K = n; g.alpha = 1.0; g.beta = 0.0;
    g.A = X1; g.lda = n; g.a_kcontig = 0; g.strideA = 0; g.a_slab = n; g.a_slab_skip = n2 - n;
    g.B = C; g.ldb = n; g.b_kcontig = 0; g.strideB = 0;
    g.C = X0; g.ldc = n; g.strideC = 0; g.batch = 1; g.cfg = 34; g.ksplit = 0;
    QTRY(dev_gemm(g));
  } else QTRY(gemm(n, n, n, 1.0, C, n, false, X1, n, false, 0.0, X0, n, np, 0, n2, n2, tcfg));
  // pair-first MO tensor Mp[P(r',s')][p'][q'] = (r's'|p'q') in X0 (X1 is free now)
  QTRY(mo_blocks_from_pair_first(n, o, X0, X1, mo_transform_work(n), out, build_Vl));
  QTRY(lap_AO2MO.close());
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// The same MO blocks from the fragment's 3-index factor B[L][P(p,q)] (naux x npair(n), embedding basis: the `bb` of
// molbe/eri_onthefly.py:141, whose product bb^T bb IS the fragment's ERI block, :143) -- north_star's "density-fitted 3-index"
// route to the fragment-MO integrals.  The factor is transformed, not the four-index tensor:
//   Lh [L][p][q']  = B[L] C                      (naux n x n x n product)
//   Lmo[L][p'][q'] = C^T Lh[L]                   (batched over L), packed over p' >= q': Lpk[L][P(p'q')]
//   S[P(r's')][P(p'q')] = sum_L Lpk[L,P(r's')] Lpk[L,P(p'q')]      -- 2 naux npair^2 x 9/16 flops (0.44e12 at n = 220, naux = 660)
//   Mp[P(r's')][p'][q'] = unpack of S            -- the pair-first tensor the four-index route ends with, to rounding
//   T [P(r's')][P][q']  = sum_L Lpk[L,P(r's')] Lh[L][P][q'],  P < nf  -- the 3/4-transformed integrals of the energies
// against 4 x 2 n^3 npair (1.81e12 executed with the triangular savings) for the four quarter transforms: cheaper while naux < ~8 n.
// X0, X1: the work buffers of mo_transform (mo_transform_work(n) doubles each).
int mo_transform_factor(int n, int o, int nf, int naux, const double* Bf, double* X0, double* X1, const double* C, MoIntegrals& out, bool build_Vl,
                        bool build_T34) {
  const int v = n - o;
  const int64_t np = (int64_t)n * (n + 1) / 2, n2 = (int64_t)n * n;
  if (naux <= 0 || !Bf) { set_error("mo_transform_factor: no factor"); return QEMB_ERR_ARG; }
  if ((int64_t)nf * n > mo_transform_work(n) / np) { set_error("mo_transform_factor: work buffer too small"); return QEMB_ERR_ARG; }
  out.n = n; out.o = o; out.v = v; out.nf = nf;
  TimerScope lap_AO2MO(TIMER_AO2MO);
  DBuf Lu, Lh, Lpk;
  QTRY(Lu.alloc((int64_t)naux * n2)); QTRY(Lh.alloc((int64_t)naux * n2)); QTRY(Lpk.alloc((int64_t)naux * np));
  QTRY(dev_unpack_tril_rows(naux, n, Bf, Lu));                                                      // B[L][p][q]
  QTRY(gemm((int64_t)naux * n, n, n, 1.0, Lu, n, true, C, n, false, 0.0, Lh, n));                  // Lh[(L,p)][q'] = sum_q B[L][p][q] C[q,q']
  QTRY(gemm(n, n, n, 1.0, C, n, false, Lh, n, false, 0.0, Lu, n, naux, 0, n2, n2));                 // Lmo[L][p'][q'] = sum_p C[p,p'] Lh[L][p][q']  (over Lu)
  QTRY(dev_pack_tril_rows(naux, n, Lu, Lpk));
  Lu.release();
  QTRY(df_pair_product(np, naux, Lpk, X1));                                                         // S in X1 (npair x npair)
  QTRY(dev_unpack_tril_rows(np, n, X1, X0));                                                        // Mp in X0
  if (nf > 0) {
    // T[P(r's')][(P,q')] for the first nf rows P of every slab of Lh: A(m,k) = Lpk[k][m], B(k,col) = Lh[k][col], col < nf n
    QTRY(gemm(np, (int64_t)nf * n, naux, 1.0, Lpk, np, false, Lh, n2, false, 0.0, X1, (int64_t)nf * n));
    QTRY(mo_three_quarter_blocks(n, o, nf, X1, (int64_t)nf * n, out, build_T34));
  }
  Lh.release(); Lpk.release();
  QTRY(mo_blocks_from_pair_first(n, o, X0, X1, mo_transform_work(n), out, build_Vl));
  QTRY(lap_AO2MO.close());
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
static void pick_xw_split(int64_t rows, int64_t oo, int64_t K, int& cfg, int& ks);
static void pick_long_k(int64_t M, int64_t N, int64_t K, int tile_m, int tile_n, int cfg_in, int& cfg, int& ks, bool many_slices);
int CcsdSolver::setup(MoIntegrals&& ints, const double* mo_energy_dev) {
  I_ = std::move(ints);
  o_ = I_.o; v_ = I_.v; nf_ = I_.nf;
  const int64_t o = o_, v = v_, nov = o * v, oo = o * o, vv = v * v, N2 = oo * vv;
  QTRY(eo_.alloc(o)); QTRY(ev_.alloc(v));
  QTRY(dev_d2d(eo_, mo_energy_dev, sizeof(double) * o));
  QTRY(dev_d2d(ev_, mo_energy_dev + o, sizeof(double) * v));
  // ---- constant tensors derived from the integral blocks
  QTRY(ovov_t_.alloc(nov * nov)); QTRY(Lovov_.alloc(nov * nov)); QTRY(Loovv_.alloc(N2)); QTRY(OVoovv_.alloc(N2));
  QTRY(Lovoo_.alloc(o * v * oo)); QTRY(W1base_.alloc(N2)); QTRY(W2base_.alloc(N2)); QTRY(Lph1_.alloc(N2));
  QTRY(oooo_p_.alloc(oo * oo));
  QTRY(perm4(ovov_t_, I_.ovov, o, v, o, v, 0, 3, 2, 1));                 // ovov_t[k,c,l,d] = ovov[k,d,l,c]
  QTRY(dcopy(nov * nov, I_.ovov, Lovov_));
  QTRY(axpby(nov * nov, -1.0, ovov_t_, 2.0, Lovov_));                    // Lovov = 2 ovov - ovov_t
  QTRY(perm4(Loovv_, Lovov_, o, v, o, v, 0, 2, 1, 3));                   // Loovv[k,l,c,d] = Lovov[k,c,l,d]
  QTRY(perm4(OVoovv_, I_.ovov, o, v, o, v, 0, 2, 1, 3));                 // OVoovv[k,l,c,d] = ovov[k,c,l,d]
  QTRY(perm4(Lovoo_, I_.ovoo, o, v, o, o, 2, 1, 0, 3, -1.0, 0.0));       // -ovoo[k,c,l,i] at [l,c,k,i]
  QTRY(axpby(o * v * oo, 2.0, I_.ovoo, 1.0, Lovoo_));                    // Lovoo[l,c,k,i] = 2 ovoo[lcki] - ovoo[kcli]
  QTRY(LovooT_.alloc(o * v * oo));
  QTRY(perm4(LovooT_, Lovoo_, o, v, o, o, 2, 3, 0, 1));                  // ... at [k,i,l,c]: Z[k,i] = sum_lc Lovoo[lcki] t1[lc] is then one row-wise matrix-vector pass
  QTRY(ovoo_ijka_.alloc(o * v * oo));
  QTRY(perm4(ovoo_ijka_, I_.ovoo, o, v, o, o, 0, 2, 3, 1));               // ovoo[i,a,j,k] at [i,j,k,a]
  QTRY(ovoo_kilc_.alloc(o * v * oo));
  QTRY(perm4(ovoo_kilc_, I_.ovoo, o, v, o, o, 2, 3, 0, 1));               // ovoo[l,c,k,i] at [k,i,l,c]
  QTRY(perm4(W1base_, I_.ovvo, o, v, v, o, 3, 2, 0, 1));                 // W1base[i,a,k,c] = ovvo[k,c,a,i]
  QTRY(perm4(W2base_, I_.oovv, o, o, v, v, 1, 2, 0, 3));                 // W2base[i,a,k,c] = oovv[k,i,a,c]
  QTRY(dcopy(N2, W2base_, Lph1_));
  QTRY(axpby(N2, 2.0, W1base_, -1.0, Lph1_));                            // Lph1 = 2 W1base - W2base
  {  // OVl[k,a,c,d] = ovvv[k,d,a,c], kept only as its (+/-) pair-packed images over (c,d): the tau-side dressing of
     // Wvvvv then contracts the SAME packed tau rows as the ladder, at half the flops of the dense o^2 x ov x v^2 product
    DBuf OVl;
    QTRY(OVl.alloc(o * v * vv));
    QTRY(perm4(OVl, I_.ovvv, o, v, v, v, 0, 2, 3, 1));
    QTRY(OVp_.alloc(nov * I_.ldp)); QTRY(OVm_.alloc(nov * I_.ldm));
    QTRY(dev_pack_pm_cols(nov, v, OVl, OVp_, I_.ldp, OVm_, I_.ldm));
  }
  // (kd|ac) is symmetric in (a,c): the pass ZC[k,i,a,c] = t1[id] ovvv[kdac] reads the block packed over that pair -- half the bytes of an
  // HBM-bound pass -- and its result is unpacked afterwards (a tenth of the bytes)
  {
    const int64_t npv = v * (v + 1) / 2;
    QTRY(ovvv_pk_.alloc(o * v * npv)); QTRY(ZCp_.alloc(oo * npv));
    QTRY(dev_pack_tril_rows(o * v, v, I_.ovvv, ovvv_pk_));
  }
  // G+-[(k,l)][P/Q(c,d)] = ovov[kcld] +- ovov[kdlc]: Woooo += ovov[kcld] tau[ijcd] then runs over the packed (c,d) pairs against the
  // packed tau rows the ladder builds anyway -- half the flops of the dense (oo) x (oo) x (vv) product
  QTRY(Gp_.alloc(oo * I_.ldp)); QTRY(Gm_.alloc(oo * I_.ldm));
  QTRY(dev_pack_pm_cols(oo, v, OVoovv_, Gp_, I_.ldp, Gm_, I_.ldm));
  QTRY(perm4(oooo_p_, I_.oooo, o, o, o, o, 1, 3, 0, 2));                 // oooo_p[i,j,k,l] = oooo[k,i,l,j]: the bare term of Woooo[k,l,i,j], row pair (i,j) first (pack_w_pm_sum)
  QTRY(ovoo_cikl_.alloc(v * oo * o));
  QTRY(perm4(ovoo_cikl_, I_.ovoo, o, v, o, o, 1, 3, 2, 0));              // ovoo[l,c,k,i] at [c,i,k,l]: O1 = t1 . ovoo then comes out as [j,(i,k,l)], one product
  // (ovvo / oovv stay resident: 2 x 128 MB at n = 220 buy two permutation passes per iteration)
  // ---- amplitudes and work space
  const int64_t na = nov + N2;
  QTRY(amp_.alloc(na)); QTRY(ampn_.alloc(na)); QTRY(diff_.alloc(na));
  for (DBuf* b : {&tau_, &T_, &Tp_, &S_, &W1_, &W2_, &W12_, &W12b_, &R_, &U_}) QTRY(b->alloc(N2));
  const int64_t NG = std::max<int64_t>(N2, oo * o * v);   // scratch also holds (o,o,v,o)-shaped temporaries
  QTRY(G2_.alloc(NG));
  {  // split-K slabs of the few-output, long-K products stay where the slices wrote them (added up by y_traces / t1_assemble)
    int cfg, ks;
    pick_long_k(v, v, oo * v, 64, 64, (v <= 256) ? 1 : -1, cfg, ks, false);
    QTRY(Fvv_.alloc((int64_t)gemm_slab_count(oo * v, ks) * vv));
    const int cw = (o <= 32) ? 21 : -1;
    pick_long_k(o, v, o * vv, 32, 128, cw, cfg, ks, true);
    const int64_t sa = gemm_slab_count(o * vv, ks);
    pick_long_k(o, v, o * v * o, 32, 128, cw, cfg, ks, true);
    const int64_t sb = gemm_slab_count(o * v * o, ks);
    QTRY(T1P_.alloc((sa + sb) * nov));
  }
  QTRY(Foo_.alloc(oo)); QTRY(Fov_.alloc(nov)); QTRY(Z_.alloc(oo)); QTRY(Y_.alloc(vv));
  QTRY(Ytmp_.alloc(vv)); QTRY(Loo_.alloc(oo)); QTRY(Lvv_.alloc(vv)); QTRY(Q_.alloc(oo)); QTRY(Wo_.alloc(oo * oo));
  QTRY(O1_.alloc(oo * oo)); QTRY(X_.alloc(oo * nov)); QTRY(scal_.alloc(8));
  {
    const int64_t npo = o * (o + 1) / 2, nmo = std::max<int64_t>(o * (o - 1) / 2, 1);
    // (results of the split-K products keep one slab per K slice: the consumers add them up)
    const int64_t npv = v * (v + 1) / 2, nmv = v * (v - 1) / 2;
    auto slabs_pair = [&](int64_t rows, int64_t cols, int64_t K, bool k_aware = false) { int cfg, ks; pick_pair_gemm(rows, cols, cfg, ks, k_aware ? K : 0); return (int64_t)gemm_slab_count(K, ks); };
    auto slabs_xw = [&](int64_t rows, int64_t K) { int cfg, ks; pick_xw_split(rows, oo, K, cfg, ks); return (int64_t)gemm_slab_count(K, ks); };
    QTRY(LTp_.alloc(npo * I_.ldp)); QTRY(LRp_.alloc(std::max(npo * I_.ldp, slabs_pair(npo, npv, I_.ldp) * npo * npv)));
    QTRY(LTm_.alloc(nmo * I_.ldm)); QTRY(LRm_.alloc(std::max(nmo * I_.ldm, slabs_pair(nmo, std::max<int64_t>(nmv, 1), I_.ldm) * nmo * std::max<int64_t>(nmv, 1))));
    QTRY(Xp_.alloc(slabs_pair(npo, nov, I_.ldp, true) * npo * nov)); QTRY(Xm_.alloc(slabs_pair(nmo, nov, I_.ldm, true) * nmo * nov));
    QTRY(Xwp_.alloc(slabs_xw(npo, I_.ldp) * npo * oo)); QTRY(Xwm_.alloc(slabs_xw(nmo, I_.ldm) * nmo * oo)); QTRY(Xw_.alloc(oo * oo));
    lwp_ = npo + (npo & 1); lwm_ = std::max<int64_t>(2, nmo + (nmo & 1));
    QTRY(WAp_.alloc(npo * lwp_)); QTRY(WAm_.alloc(nmo * lwm_)); QTRY(HRp_.alloc(npo * I_.ldp)); QTRY(HRm_.alloc(nmo * I_.ldm));
    QTRY(ZB_.alloc(N2)); QTRY(ZC_.alloc(N2));
  }
  first_ = true;
  return 0;
}

int CcsdSolver::make_tau(const double* t1, const double* t2, double* tau) {
  const int64_t o = o_, v = v_;
  Outer4Desc d{};
  d.dim[0] = o; d.dim[1] = o; d.dim[2] = v; d.dim[3] = v;
  d.u = t1; d.su0 = v; d.su2 = 1; d.v = t1; d.sv1 = v; d.sv3 = 1;
  d.out = tau; d.so[0] = o * v * v; d.so[1] = v * v; d.so[2] = v; d.so[3] = 1;
  d.alpha = 1.0; d.beta = 1.0; d.base = t2;          // tau = t2 + t1 (x) t1 in one pass
  return dev_outer4(d);
}

int CcsdSolver::energy(const double* t1, const double* t2, double* e) {
  // E = sum (2 ovov[iajb] - ovov[ibja]) tau[ijab] = <Loovv, tau>   (f_ov = 0)
  QTRY(make_tau(t1, t2, tau_));
  QTRY(dev_dot((int64_t)o_ * o_ * v_ * v_, Loovv_, tau_, scal_));
  if (e) QTRY(dev_d2h(e, scal_, sizeof(double)));      // (null: the value stays in scal_[0] -- fetch_energy)
  return 0;
}
int CcsdSolver::fetch_energy() { return dev_d2h(&ecc_, scal_, sizeof(double)); }

int CcsdSolver::init_amps(bool defer_energy) {
  const int64_t o = o_, v = v_;
  QTRY(dev_fill(t1(), o * v, 0.0));                                      // t1 = f_ov / e_ia = 0
  QTRY(dcopy(o * o * v * v, OVoovv_, t2()));                             // t2 = ovov[i,a,j,b] / e_ijab
  QTRY(dev_div_denom(t2(), o, o, v, v, eo_, eo_, ev_, ev_));
  first_ = true;
  diis_.clear();
  return energy(t1(), t2(), defer_energy ? nullptr : &ecc_);
}

int CcsdSolver::set_amps(const double* t1d, const double* t2d, bool defer_energy) {
  QTRY(dev_d2d(t1(), t1d, sizeof(double) * o_ * v_));
  QTRY(dev_d2d(t2(), t2d, sizeof(double) * (int64_t)o_ * o_ * v_ * v_));
  first_ = true;
  diis_.clear();
  return energy(t1(), t2(), defer_energy ? nullptr : &ecc_);
}

// tile configuration and K split for the "few packed pair rows x many columns" GEMMs (ladder, tau-side dressing)
void pick_pair_gemm(int64_t rows, int64_t cols, int& cfg, int& ks, int64_t K) {
  cfg = -1; ks = 0;
  if (cols < 1024) return;
  // the row-tile height with the least ESTIMATED TIME among the configurations that exist: padded rows weighted by what a row costs on
  // that tile relative to the 7 x 2 / 6 x 2 wave tiles (the single-column wave tiles 11 / 12 read 8 / 5 LDS fragments per 7 / 4 MFMAs and
  // run at ~0.8 / ~0.6 of their rate on these long-K products; profiles/r01_gemm_microbench_ladder_tiles.jsonl).  n_occ = 20 gets the
  // 224-row tile for its 210 symmetric and the 192-row tile for its 190 antisymmetric pairs; n_occ = 30 three 160-row tiles for 465;
  // n_occ = 40 (820 rows) four 224-row tiles, not thirteen 64-row ones.  (The dispatcher's own choice for "few tiles" would be the
  // 64 x 64 tile, a quarter of the rate.)
  static const struct { int rows, cfg, cols; double cost; } cand[] = {{224, 13, 128, 1.0}, {192, 15, 128, 1.0}, {160, 35, 128, 1.03}, {128, 4, 256, 1.0},
                                                                      {112, 11, 128, 1.25}, {80, 36, 128, 1.08}, {64, 12, 128, 1.6}, {48, 38, 128, 1.12}};
  double best_cost = -1.0;
  int64_t tiles = 0;
  for (const auto& c : cand) {
    const int64_t mt = (rows + c.rows - 1) / c.rows;
    if (c.cfg == 38 && mt > 1) continue;      // (the 48-row tile only where it holds all rows: n_occ <= 9)
    const double cost = (double)(mt * c.rows) * c.cost;
    if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; cfg = c.cfg; tiles = mt * ((cols + c.cols - 1) / c.cols); }
  }
  double best = 0.0;
  for (int c = 1; c <= 8; ++c) {
    const int64_t units = tiles * c, rounds = (units + 255) / 256;
    const double eff = (double)units / (double)(rounds * 256) - (c == 1 ? 0.0 : 0.002 * c);
    if (units >= 512 && eff > best + 1e-9) { best = eff; ks = c; }
  }
  if (ks == 0) ks = (int)std::max<int64_t>(1, std::min<int64_t>(8, (1024 + tiles - 1) / tiles));
  // a product whose eight slices still leave most CUs without a workgroup and whose K is long (the tau-side dressing of mid-size fragments: 12 column tiles x 8 at
  // n = 132, K = 7260 -- 85 us at 19 TFLOP/s): more, shorter slices, down to ~256 k each (K = 0: the caller did not say, the rule above stands)
  if (K > 0 && tiles * ks < 256) ks = (int)std::max<int64_t>(ks, std::min<int64_t>(std::min<int64_t>(32, K / 256), (512 + tiles - 1) / tiles));
}

// C = A B^T (both operands K-contiguous) whose consumer adds the split-K slabs itself: with ks > 1 the slices' partial products stay in slabs
// [S][M][N] at C (no reduction pass), else C is the plain product with leading dimension ldc
struct SlabGemm { int S = 1; int64_t stride = 0, ld = 0; };
static int gemm_slabs(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int cfg, int ks, SlabGemm& out,
                      bool a_kc = true, bool b_kc = true) {
  GemmDesc g{};
  g.M = M; g.N = N; g.K = K; g.alpha = 1.0; g.beta = 0.0;
  g.A = A; g.lda = lda; g.a_kcontig = a_kc ? 1 : 0; g.B = B; g.ldb = ldb; g.b_kcontig = b_kc ? 1 : 0;
  g.C = C; g.ldc = ldc; g.batch = 1; g.cfg = cfg; g.ksplit = ks;
  out.S = gemm_slab_count(K, ks);
  if (out.S > 1) { g.keep_slabs = 1; out.stride = M * N; out.ld = N; }
  else { out.stride = 0; out.ld = ldc; }
  return dev_gemm(g);
}
// tile configuration and K split of the Xw products (output only npair(o) x o^2, K = packed virtual pairs): 64 x 64 tiles and enough K slices for
// ~2 workgroups per CU
static void pick_xw_split(int64_t rows, int64_t oo, int64_t K, int& cfg, int& ks) {
  cfg = -1; ks = 0;
  if (K < 2048) return;
  const int64_t tiles = ((rows + 63) / 64) * ((oo + 63) / 64);
  cfg = 1;
  ks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, K / 256), (512 + tiles - 1) / tiles));
}
// The K split dev_gemm would pick by itself for a product with `tiles` output tiles (gemm_f64.hip launch_cfg), spelled out so that the caller can keep the slabs
static int auto_ksplit(int64_t tiles, int64_t K) {
  if (tiles >= 256 || K < 1024) return 0;
  int64_t S = (768 + tiles - 1) / tiles;
  if (S > K / 256) S = K / 256;
  return S > 1 ? (int)S : 0;
}
// few-output, long-K products (Fvv', the two ovvv / ovoo terms of the T1 equation): tile configuration and K split; cfg < 0: the dispatcher's own choice, no slabs
static void pick_long_k(int64_t M, int64_t N, int64_t K, int tile_m, int tile_n, int cfg_in, int& cfg, int& ks, bool many_slices = false) {
  cfg = cfg_in; ks = 0;
  if (cfg_in < 0) return;
  ks = auto_ksplit(((M + tile_m - 1) / tile_m) * ((N + tile_n - 1) / tile_n), K);
  // a result of one tile (small fragments) whose consumer adds the slabs with all its threads (ccsd_t1_assemble): slices of ~64-96 k instead of ~288 -- a
  // workgroup then runs 4-6 k-steps instead of 18, and there are hundreds of them (octane: 20 + 31 -> ~10 us each for the two T1 products).  (A scalar-FMA
  // kernel for these shapes -- one workgroup per slice out of LDS, no padded MFMA tile -- was measured slower than the tiled kernel on the same slices.)
  if (many_slices && ks > 1 && M <= 32 && N <= 32) ks = (int)std::min<int64_t>(512, K / 64);
}
// pp-ladder through the (+/-) pair-packed operands (see the comment in update_amps)
int CcsdSolver::apply_ladder(const double* x, double* out, bool rows_packed, bool hh) {
  const int64_t o = o_, v = v_;
  {
    const int64_t npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2, npv = v * (v + 1) / 2, nmv = v * (v - 1) / 2;
    const int64_t ldp = I_.ldp, ldm = I_.ldm;
    if (!rows_packed) QTRY(dev_ladder_pack_tau(o, v, x, LTp_, ldp, LTm_, ldm));
    int cfg, ks;
    TimerScope lap_LADDER(TIMER_LADDER);
    // (split over K, the slices' partial products stay in slabs that the scatter below adds up: no reduction pass over the packed results)
    SlabGemm sp, sm;
    pick_pair_gemm(npo, npv, cfg, ks);
    if (cfg == 13 || cfg == 15) cfg += 10;      // same tiles under the ladder's own kernel symbol (profiles)
    QTRY(dev_region_begin());
    QTRY(gemm_slabs(npo, npv, ldp, LTp_, ldp, I_.Vp, ldp, LRp_, ldp, cfg, ks, sp));
    QTRY(dev_region_chain());
    if (nmo > 0 && nmv > 0) {
      pick_pair_gemm(nmo, nmv, cfg, ks);
      if (cfg == 13 || cfg == 15) cfg += 10;
      QTRY(gemm_slabs(nmo, nmv, ldm, LTm_, ldm, I_.Vm, ldm, LRm_, ldm, cfg, ks, sm));
    }
    QTRY(dev_region_end());
    QTRY(lap_LADDER.close());
    if (hh) {
      // hole-hole ladder on the same packed rows, now as the RIGHT operand (K = packed occupied pairs):
      //   HR+[P(ij),P(ab)] = sum_{k>=l} WA+[P(ij),P(kl)] LTp[P(kl),P(ab)],   HR-[Q(ij),Q(ab)] = sum_{k>l} WA-[Q(ij),Q(kl)] LTm[Q(kl),Q(ab)]
      // -- a quarter of the flops of the dense o^2 x v^2 x o^2 product; the scatter adds them to the pp-ladder rows (doubling the a = b
      // columns of HR+: LTp carries 1/2 there) and writes the sum as the first contribution to `out`
      static const struct { int rows, cfg; } cand[] = {{224, 13}, {192, 15}, {160, 35}, {128, 0}, {64, 1}};
      auto tile_for = [&](int64_t rows) { int best = -1; int64_t pad = -1; for (const auto& c : cand) { const int64_t q = (rows + c.rows - 1) / c.rows * c.rows; if (pad < 0 || q < pad) { pad = q; best = c.cfg; } } return best; };
      QTRY(dev_region_begin());
      QTRY(gemm(npo, npv, npo, 1.0, WAp_, lwp_, true, LTp_, ldp, false, 0.0, HRp_, ldp, 1, 0, 0, 0, npv >= 2048 ? tile_for(npo) : -1));
      QTRY(dev_region_chain());
      if (nmo > 0 && nmv > 0) QTRY(gemm(nmo, nmv, nmo, 1.0, WAm_, lwm_, true, LTm_, ldm, false, 0.0, HRm_, ldm, 1, 0, 0, 0, nmv >= 2048 ? tile_for(nmo) : -1));
      QTRY(dev_region_end());
      QTRY(dev_ladder_scatter_pm2(o, v, LRp_, sp.ld, LRm_, sm.ld, HRp_, (nmo > 0 && nmv > 0) ? HRm_.p : nullptr, 1, out, sp.S, sp.stride, sm.S, sm.stride, ldp, ldm));
    } else {
      QTRY(dev_ladder_scatter_pm2(o, v, LRp_, sp.ld, LRm_, sm.ld, nullptr, nullptr, 0, out, sp.S, sp.stride, sm.S, sm.stride));
    }
  }
  return 0;
}

int CcsdSolver::update_amps(double* t1n, double* t2n) {
  const int64_t o = o_, v = v_, nov = o * v, oo = o * o, vv = v * v;
  const double* t1 = this->t1();
  const double* t2 = this->t2();
  // Parallel regions (dev_region_begin / chain / end, round 5): operations that stand NEXT to each other here and do not depend on each other are
  // marked as the chains of a region; the lock step of small fragments (dev_tape_run) then issues the same kernel of both chains and of all fragments as
  // one launch.  Outside a tape capture the calls do nothing.  The order of the operations is the producer-next-to-consumer order of rounds 1-4: a version that
  // reordered the update into seven wide regions was 0.5-0.8 ms per octane BE2 sweep slower on the same box (operands left the L2 between producer and
  // consumer) and was withdrawn; marking only neighbours costs nothing on BE2 and gains ~0.7 ms per BE3 sweep.
  // ---- amplitude layouts.  tau_ already holds tau(t1, t2): every change of the amplitudes (init_amps, set_amps, iterate) ends
  // with energy(), which builds it.
  // T [k,c,j,b] = t2[k,j,c,b], Tp[k,c,j,b] = t2[k,j,b,c], S = u = 2T - Tp (Theta_ph), and the t1-dressed ring operands
  // u~ = u - 2 t1(x)t1 (W12_), Tp~ = Tp + 2 t1(x)t1 (W12b_), t1(x)t1[(ia),(ld)] = t1[id] t1[la] -- one pass over t2
  QTRY(dev_region_begin());
  QTRY(dev_ccsd_ph_layouts(o, v, t2, t1, T_, Tp_, S_, W12_, W12b_, R_));   // R_: Th[i,k,d,c] = 2 t2[ikcd] - t2[ikdc], scratch until the rings
  QTRY(dev_region_chain());
  QTRY(dev_ladder_pack_tau(o, v, tau_, LTp_, I_.ldp, LTm_, I_.ldm));     // the packed tau rows: also read by the ladder and by the tau-side dressing below
  QTRY(dev_region_end());

  // ---- one- and two-index intermediates (energy-shifted: Foo - eps, Fvv - eps, ...)
  {  // Xw[i,j,k,l] = ovov[kcld] tau[ijcd] (the quadratic part of Woooo, added to it below) over the (+/-) packed (c,d) pairs:  X[ij] = Xp + Xm, X[ji] = Xp - Xm (i > j),
     //   Xp[P(ij),(kl)] = sum_{c>=d} LTp[P(ij),P(cd)] G+[(kl),P(cd)],  Xm[Q(ij),(kl)] = sum_{c>d} LTm G-   (LTp carries 1/2 on c = d, G+ is doubled there)
    const int64_t npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2, nmv = v * (v - 1) / 2;
    auto split = [&](int64_t rows, int64_t K, int& cfg, int& ks) { pick_xw_split(rows, oo, K, cfg, ks); };
    int cfg, ks;
    SlabGemm sp, sm;
    split(npo, I_.ldp, cfg, ks);
    QTRY(dev_region_begin());
    QTRY(gemm_slabs(npo, oo, I_.ldp, LTp_, I_.ldp, Gp_, I_.ldp, Xwp_, oo, cfg, ks, sp));
    QTRY(dev_region_chain());
    if (nmo > 0) {
      if (nmv > 0) { split(nmo, I_.ldm, cfg, ks); QTRY(gemm_slabs(nmo, oo, I_.ldm, LTm_, I_.ldm, Gm_, I_.ldm, Xwm_, oo, cfg, ks, sm)); }
      else QTRY(dev_fill(Xwm_, nmo * oo, 0.0));
    }
    QTRY(dev_region_end());
    QTRY(dev_scatter_pm_rows(o, oo, Xwp_, Xwm_, Xw_, nullptr, sp.S, sp.stride, sm.S, sm.stride));      // Xw[i,j,k,l] (the K slices' slabs added on the way)
  }
  // (Foo' and Z only ever enter as Loo' = Foo' + Z, Fvv' and Y as Lvv' = Fvv' + Y: each pair is accumulated in place)
  // Foo'[k,i] = sum_{lcd} (2 ovov[kcld] - ovov[kdlc]) tau[ilcd] = sum_l (2 Xw[i,l,k,l] - Xw[l,i,k,l]): a partial trace of Xw instead of
  // a pass over two o^2 v^2 tensors
  QTRY(dev_foo_from_x(o, Xw_, Loo_));
  // Fvv'[a,c] = -sum tau[klxa] Loovv[klxc]  (64 x 64 tiles: split-K supplies the blocks; the slabs are added up -- with the sign -- by the pass that forms Lvv' below)
  SlabGemm sfvv;
  double fvv_scale = -1.0;
  {
    int cfg, ks;
    pick_long_k(v, v, oo * v, 64, 64, (v <= 256) ? 1 : -1, cfg, ks);
    if (ks > 1) QTRY(gemm_slabs(v, v, oo * v, tau_, v, Loovv_, v, Fvv_, v, cfg, ks, sfvv, false, false));
    else { QTRY(gemm(v, v, oo * v, -1.0, tau_, v, false, Loovv_, v, false, 0.0, Fvv_, v, 1, 0, 0, 0, cfg)); fvv_scale = 1.0; }
  }
  // Fov[k,c] = Lovov[(kc),:] . t1  and  Loo' = Foo' + Z[k,i], Z = LovooT[(ki),:] . t1: two matrix-vector passes, one launch
  QTRY(dev_gemv_rows_two(nov, nov, Lovov_, nov, t1, Fov_, 1.0, 0.0, oo, nov, LovooT_, nov, t1, Loo_, 1.0, 1.0));
  // The two t1-contractions of ovvv are formed ONCE per iteration (each is one pass over the 1.28 GB block) and serve the
  // ring intermediates, the X1 term and -- through their k = i traces -- the Y intermediate:
  // (n_occ <= 32 columns / rows: 128 x 32 and 32 x 128 tiles instead of padding n_occ to a 64-wide tile, which made these
  //  HBM-bound passes MFMA-bound)
  const int cfg_tall = (o <= 32) ? 20 : -1, cfg_wide = (o <= 32) ? 21 : -1;
  QTRY(dev_region_begin());
  QTRY(gemm(o * vv, o, v, 1.0, I_.ovvv, v, true, t1, v, true, 0.0, ZB_, o, 1, 0, 0, 0, cfg_tall));     // ZB[k,c,a,i] = ovvv[kcad] t1[id]
  QTRY(dev_region_chain());
  {  // ZC[k,i,a,c] = t1[id] ovvv[kdac], symmetric in (a,c): on the pair-packed block, then unpacked
    const int64_t npv = v * (v + 1) / 2;
    const bool vec_ok = (npv % 2) == 0;     // (the 32 x 128 tile wants 16-byte aligned rows; odd npair(v): the dispatcher's choice)
    QTRY(gemm(o, npv, v, 1.0, t1, v, true, ovvv_pk_, npv, false, 0.0, ZCp_, npv, o, 0, v * npv, o * npv, vec_ok ? cfg_wide : -1));
    QTRY(dev_unpack_tril_rows(oo, v, ZCp_, ZC_));
  }
  QTRY(dev_region_end());
  QTRY(dev_ccsd_y_traces(o, v, ZC_, ZB_, Lvv_, Fvv_, sfvv.S, sfvv.stride, fvv_scale));                               // Lvv' = Fvv' + Y,  Y[a,c] = 2 sum_k ZC[k,k,a,c] - sum_k ZB[k,c,a,k]

  // ---- T1 equation
  // The two long-K terms first, as the slabs their K slices leave:  PA[i,a] = (2 ovvv[kdac] - ovvv[kcad]) t2[ikcd],  PB[i,a] = (2 ovoo[lcki] - ovoo[kcli]) t2[klac]
  SlabGemm spa, spb;
  double* PA = T1P_.p;
  QTRY(dev_region_begin());
  {
    int cfg, ks;
    pick_long_k(o, v, o * vv, 32, 128, cfg_wide, cfg, ks, true);
    if (ks > 1) QTRY(gemm_slabs(o, v, o * vv, R_, o * vv, I_.ovvv, v, PA, v, cfg, ks, spa, true, false));
    else { QTRY(gemm(o, v, o * vv, 1.0, R_, o * vv, true, I_.ovvv, v, false, 0.0, PA, v, 1, 0, 0, 0, cfg)); spa.S = 1; }
  }
  double* PB = PA + (int64_t)spa.S * nov;
  QTRY(dev_region_chain());
  {
    int cfg, ks;
    pick_long_k(o, v, o * v * o, 32, 128, cfg_wide, cfg, ks, true);
    if (ks > 1) QTRY(gemm_slabs(o, v, o * v * o, Lovoo_, o, T_, v, PB, v, cfg, ks, spb, false, false));
    else { QTRY(gemm(o, v, o * v * o, 1.0, Lovoo_, o, false, T_, v, false, 0.0, PB, v, 1, 0, 0, 0, cfg)); spb.S = 1; }
  }
  QTRY(dev_region_end());
  // t1n = (Fvv'+Y)_ac t1[ic] - (Foo'+Z)_ki t1[ka] + Fov_kc t1[ic] t1[ka] + Fov_kc (2 t2[kica] - t2[ikca]) + (2 ovvo[kcai] - oovv[kiac]) t1[kc] + PA - PB:
  // the small products, the two passes over o^2 v^2 operands and the slab sums in one launch (a workgroup per element)
  QTRY(dev_ccsd_t1_assemble(o, v, t1, Lvv_, Loo_, Fov_, S_, Lph1_, PA, spa.S, nov, PB, spb.S, nov, t1n));

  // ---- T2 equation: direct (unsymmetrised) part
  // (the bare ovov[i,a,j,b] term is added by the finishing kernel)
  // Woooo[k,l,i,j]
  //   = oooo[kilj] + ovov[kcld] tau[ijcd] (Xw[i,j,k,l], formed at the top) + ovoo[lcki] t1[jc] (O1[j,i,k,l]) + ovoo[kclj] t1[ic] (O1[i,j,l,k])
  // is never stored: its (+/-) pair-packed images -- Woooo[klij] tau[klab] goes with the ladder below, through packed pairs -- are formed from the four terms
  QTRY(gemm(o, oo * o, v, 1.0, t1, v, true, ovoo_cikl_, oo * o, false, 0.0, O1_, oo * o));          // O1[j,i,k,l] = sum_c t1[j,c] ovoo[l,c,k,i]
  QTRY(dev_pack_w_pm_sum(o, oooo_p_, Xw_, O1_, WAp_, lwp_, WAm_, lwm_));
  // pp-ladder (the dominant kernel)
  // R_ijab = sum_cd (ac|bd) tau_ijcd through pair-packed symmetric / antisymmetric combinations:
  //   R = R+ + R-,  R+[P(ij),P(ab)] = sum_{c>=d} Vp[P(ab),P(cd)] Tp[P(ij),P(cd)],  R-[Q(ij),Q(ab)] = sum_{c>d} Vm Tm,
  // using tau[j,i,d,c] = tau[i,j,c,d] and (ac|bd) = (bd|ac): only i >= j rows, a >= b columns and c >= d contractions
  // are computed -- 2 npair(o) npair(v)^2 + 2 npair'(o) npair'(v)^2 flops = 1/4 of the dense 2 o^2 v^4 -- and the
  // operands Vp, Vm (6.4 GB together at v = 200) are each streamed ONCE through a tile that holds every packed (ij) row
  // (224 x 128, 8 waves), K split over workgroups to fill whole rounds of the 256 CUs.
  QTRY(apply_ladder(tau_, t2n, /*rows_packed=*/true, /*hh=*/true));               // (first writer of t2n)

  // ---- T2 equation: terms that enter as P(X) accumulate in U
  // Lvv'[a,c] t2[ijcb] enters as its P-partner t2[ijac] Lvv'[b,c] (U is only used as U + U^T(ji,ba)): ONE (o^2 v) x v x v product
  QTRY(gemm(oo * v, v, v, 1.0, t2, v, true, Lvv_, v, true, 0.0, U_, v));
  QTRY(gemm(o, o * vv, o, -1.0, Loo_, o, false, t2, o * vv, false, 1.0, U_, o * vv, 1, 0, 0, 0, cfg_wide));   // -Loo'[k,i] t2[kjab]
  //   t1-dressing of Wvvvv folded on the tau side: -t1[kb] (tau[ijcd] ovvv[kdac])
  {  // X[i,j,k,a] = tau[ijcd] OVl[k,a,c,d] from the packed tau rows LTp/LTm that apply_ladder just built:
     //   X[ij] = Xp + Xm, X[ji] = Xp - Xm (i > j),  Xp = LTp OVp^T (c >= d),  Xm = LTm OVm^T (c > d)
    const int64_t npo = o * (o + 1) / 2, nmo = o * (o - 1) / 2;
    int cfg, ks;
    SlabGemm sp, sm;
    pick_pair_gemm(npo, nov, cfg, ks, I_.ldp);
    QTRY(dev_region_begin());
    QTRY(gemm_slabs(npo, nov, I_.ldp, LTp_, I_.ldp, OVp_, I_.ldp, Xp_, nov, cfg, ks, sp));
    QTRY(dev_region_chain());
    if (nmo > 0) {
      pick_pair_gemm(nmo, nov, cfg, ks, I_.ldm);
      QTRY(gemm_slabs(nmo, nov, I_.ldm, LTm_, I_.ldm, OVm_, I_.ldm, Xm_, nov, cfg, ks, sm));
    }
    QTRY(dev_region_end());
    QTRY(dev_scatter_pm_rows(o, nov, Xp_, Xm_, X_, ovoo_ijka_, sp.S, sp.stride, sm.S, sm.stride));   // ... + ovoo[i,a,j,k]: the second term of A below, added on the way
  }
  //   X1 = (ovvv[iacb] - oovv[kibc] t1[ka]) t1[jc],   X2 = (ovvo[kcai] t1[jc] + ovoo[iajk]) t1[kb] (enters with a minus sign)
  QTRY(dev_region_begin());
  QTRY(perm4(U_, ZB_, o, v, v, o, 0, 3, 1, 2, 1.0, 1.0));                          // U[i,j,a,b] += t1[jc] ovvv[i,a,b,c] = ZB[i,a,b,j]
  QTRY(dev_region_chain());
  // Every other t1-dressing term has the form U[i,j,a,b] -= sum_k A[i,j,k,a] t1[k,b] (the oovv one through its P-partner), with
  // A only o^3 v large: the A's are summed first and ONE rank-n_occ update passes over U.
  //   A = X[i,j,k,a] + ovoo[i,a,j,k] + (oovv[(k,j,a),c] t1[i,c]) + (t1[j,c] ovvo[k,c,a,i])
  QTRY(gemm(oo * v, o, v, 1.0, I_.oovv, v, true, t1, v, true, 0.0, G2_, o, 1, 0, 0, 0, cfg_tall));                                   // G2[k,j,a,i] = oovv[(kja),c] t1[ic]
  QTRY(gemm(o, v * o, v, 1.0, t1, v, true, I_.ovvo, v * o, false, 1.0, G2_, v * o, o, 0, v * v * o, o * v * o, cfg_wide));          //          += t1[jc] ovvo[k,c,a,i]
  QTRY(dev_region_end());
  QTRY(perm4(X_, G2_, o, o, v, o, 3, 1, 0, 2, 1.0, 1.0));                          // A[i,j,k,a] += G2[k,j,a,i]
  QTRY(dev_small_k_update(oo, v, v, o, -1.0, X_, nov, t1, 0, U_, vv));                 // U[ij][a][b] -= sum_k A[ij][k][a] t1[k][b]
  // ---- ph rings
  TimerScope lap_RINGS(TIMER_RINGS);
  // The t2-dependent parts of both ring intermediates come from TWO (ov)^3 products instead of three: with
  //   u~ = 2T - Tp - 2 t1(x)t1,  Tp~ = Tp + 2 t1(x)t1   (t1(x)t1[(ia),(ld)] = t1[id] t1[la]),  L = 2 ovov - ovov_t,
  //   Wvoov += 1/4 u~ L - 1/4 Tp~ ovov_t,      Wvovo -= 1/2 Tp~ ovov_t
  // (expand: 1/4 (2T - Tp)(2 ovov - ovov_t) = (T - Tp/2) ovov - T ovov_t / 2 + Tp ovov_t / 4, and the t1(x)t1 pieces
  // reproduce -ovov[ldkc] t1[id] t1[la] and -ovov[lckd] t1[id] t1[la]).
  // (S = u, W12_ = u~ and W12b_ = Tp~ were formed by dev_ccsd_ph_layouts at the top of the update)
  // the rank-n_occ pieces -ovoo[kcli] t1[la] (Wvoov) and -t1[la] ovoo[lcki] (Wvovo) are subtracted from ZB / ZC in place (their last
  // readers -- the Y traces and the X1 term -- are done), so ONE transposing pass per intermediate carries both pieces
  QTRY(dev_small_k_update(nov, v, o, o, -1.0, t1, 0, I_.ovoo, oo, ZB_, v * o));    // ZB[k,c,a,i] -= sum_l t1[l,a] ovoo[k,c,l,i]
  QTRY(perm4(W1_, ZB_, o, v, v, o, 3, 2, 0, 1, 1.0, 1.0, W1base_));                // W1 = W1base + ovvv[kcad] t1[id] - ovoo[kcli] t1[la]
  // The right-hand operands of all four (ov)^3 products are symmetric matrices over (kc),(ld) -- L and ovov_t by the integral symmetry
  // (kc|ld) = (ld|kc), T' and u by t2[k,j,c,b] = t2[j,k,b,c] -- so each is passed in its K-contiguous (transposed) reading: both operands
  // of the GEMM are then staged through the conflict-free [row][BK+2] LDS image, on the 128 x 256 tile (512 tiles at ov = 4000: two per CU).
  // (small fragments in a lock-step sweep, QEMB_LOCKSTEP_PEERS=1: `peers` products of this shape run in ONE grouped launch, enough to fill the chip with
  //  64 x 64 tiles.  Measured on six octane fragments, o v = 441: 294 such workgroups on 256 CUs are SLOWER than the 1176 of the 32 x 32 tile the
  //  dispatcher picks for a lone product -- 11.3 against 11.0 ms per sweep -- so the hint stays off.)
  const int64_t ring_tiles64 = ((nov + 63) / 64) * ((nov + 63) / 64) * dev_gemm_peers();
  // (1024 <= o v < 2048, one mid-size fragment on the chip: 96 x 96 tiles with K in two slices -- 15 x 15 x 2 workgroups at o v = 1440, 119 us against the 151 us
  //  of 23 x 23 tiles of 64 x 64; tools/mid_gemm_bench.py.  QEMB_RING96=0: the 64 x 64 tiles, for A/B runs)
  static const bool ring96 = !(std::getenv("QEMB_RING96") && std::getenv("QEMB_RING96")[0] == '0');
  const bool mid_ring = ring96 && nov >= 1024 && nov < 2048 && dev_gemm_peers() == 1;
  const int cfg_ring = (nov >= 2048) ? 4 : mid_ring ? 37 : (nov >= 256 && ring_tiles64 >= 200) ? 1 : -1;
  auto ring = [&](double al, const double* A, const double* Bsym, double be, double* C) {
    return gemm(nov, nov, nov, al, A, nov, true, Bsym, nov, true, be, C, nov, 1, 0, 0, 0, cfg_ring, mid_ring ? 2 : 0);
  };
  QTRY(ring(0.25, W12_, Lovov_, 1.0, W1_));                                        // + 1/4 u~ L
  //   W2[(ia),(kc)] = Wvovo[a,k,c,i]
  QTRY(dev_small_k_update(oo, v, v, o, -1.0, t1, 0, ovoo_kilc_, nov, ZC_, vv));     // ZC[k,i,a,c] -= sum_l t1[l,a] ovoo[l,c,k,i]
  // W2 = W2base + t1[id] ovvv[kdac] - t1[la] ovoo[lcki].  The Tp~ ovov_t product enters Wvoov with -1/4 and Wvovo with -1/2, so it cancels in
  // Wvoov - Wvovo/2: that combination (R) is formed first -- as a second output of the pass that writes W2 -- then the GEMM accumulates the product
  // straight into Wvovo.
  {
    Copy4Desc c{};
    const int64_t d[4] = {o, o, v, v};       // ZC[k,i,a,c] -> [i,a,k,c]
    const int perm[4] = {1, 2, 0, 3};
    int64_t od[4], ostr[4];
    for (int k = 0; k < 4; ++k) od[k] = d[perm[k]];
    ostr[3] = 1; ostr[2] = od[3]; ostr[1] = od[3] * od[2]; ostr[0] = od[3] * od[2] * od[1];
    c.in = ZC_; c.out = W2_; c.alpha = 1.0; c.beta = 1.0; c.base = W2base_;
    c.si[3] = 1; c.si[2] = d[3]; c.si[1] = d[3] * d[2]; c.si[0] = d[3] * d[2] * d[1];
    for (int k = 0; k < 4; ++k) { c.dim[k] = d[k]; c.so[perm[k]] = ostr[k]; }
    c.out2 = R_; c.in2 = W1_; c.c2a = 1.0; c.c2b = -0.5;                            // R = Wvoov - Wvovo/2
    QTRY(dev_copy4(c));
  }
  QTRY(ring(-0.5, W12b_, ovov_t_, 1.0, W2_));                                       // Wvovo -= 1/2 Tp~ ovov_t
  // Update, also two products:  (2 Wvoov - Wvovo) T - Wvoov Tp = (Wvoov - Wvovo/2) u - (Wvovo Tp)/2  with T = (u + Tp)/2
  // The two products stay where the GEMMs leave them ([i,a,j,b]); the finishing pass takes U[i,j,a,b] += RS[i,a,j,b] - A3[i,a,j,b] / 2 - A3[i,b,j,a] from there
  QTRY(dev_region_begin());
  QTRY(ring(1.0, W2_, Tp_, 0.0, W1_));                                             // A3 = Wvovo[bkci] t2[kjac] at W1[i,b,j,a]
  QTRY(dev_region_chain());
  QTRY(ring(1.0, R_, S_, 0.0, W12_));                                              // RS = (Wvoov - Wvovo/2) u   (W12_: its last reader was the first product)
  QTRY(dev_region_end());
  QTRY(lap_RINGS.close());

  // ---- symmetrise and divide: (t2n + ovov + U' + U'^T(ji,ba)) / D in one pass over each (i >= j) pair of tiles, t1n / D on the way
  QTRY(dev_ccsd_finish_t2_rings(o, v, t2n, U_, OVoovv_, W12_, W1_, eo_, ev_, t1n));
  return 0;
}

int CcsdSolver::iterate(double* e_corr, double* normt) {
  TimerScope lap_ITER(TIMER_ITER);
  QTRY(iterate_update(false, false, nullptr));
  QTRY(iterate_post(e_corr, normt));
  QTRY(lap_ITER.close());
  return 0;
}

// The amplitude update of one iteration.  Launch-bound regime (small fragments): update_amps is recorded once (after one eager pass has
// settled every workspace) and replayed -- as an executable hipGraph, or, in the lock-step sweep over several fragments (prefer_tape), as
// a TAPE that dev_tape_run executes together with the other fragments' tapes.  defer_tape: when the tape exists, do not run it -- the
// caller runs it grouped (*deferred = true).  Large fragments are GEMM bound and keep the eager path with its per-kernel timers.
// (hipGraph replay under rocprofv3's kernel tracing aborts inside the profiler on this ROCm: fall back to eager launches there)
int CcsdSolver::iterate_update(bool prefer_tape, bool defer_tape, bool* deferred) {
  static const bool graphs_enabled = [] {
    const char* e = std::getenv("QEMB_GRAPH");
    if (e) return e[0] != '0';
    const char* pre = std::getenv("LD_PRELOAD");
    if (std::getenv("ROCP_TOOL_LIBRARIES") || (pre && std::strstr(pre, "rocprofiler"))) return false;
    return true;
  }();
  static const bool tape_replay = std::getenv("QEMB_TAPE") != nullptr;      // diagnostic: replay the captured sequence through dev_tape_run (one tape)
  if (deferred) *deferred = false;
  const bool small = (int64_t)o_ * o_ * v_ * v_ <= (int64_t)1 << 22;
  // With DIIS the new amplitudes and their error vector go straight into the storage of the next DIIS slot (no staging copies);
  // the replay of small fragments has its output address baked in and keeps the staging buffer.
  const bool use_diis = !first_ && !diis_.empty();
  const bool replayable = graphs_enabled && small && graph_ok_;
  double* out = (use_diis && !replayable) ? diis_[0].next_x() : ampn_.p;
  last_out_ = out; last_use_diis_ = use_diis; last_replayable_ = replayable;
  if (replayable && tape_) {
    if (defer_tape && deferred) { *deferred = true; return 0; }
    QTRY(dev_tape_run(&tape_, 1));
  } else if (replayable && graph_) {
    QTRY(dev_graph_launch(graph_));
  } else if (replayable && eager_iters_ >= 1) {
    const int rc = dev_graph_begin((prefer_tape || tape_replay) ? 1 : 0);
    if (rc == 0 && (prefer_tape || tape_replay)) {
      const int rc2 = update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_);
      dev_tape_t t = nullptr;
      const int rc3 = dev_tape_end(&t);
      if (rc2 || rc3) { graph_ok_ = false; if (t) dev_tape_destroy(t); if (rc2) return rc2; if (rc3 < 0) return rc3; QTRY(update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_)); }
      else { tape_ = t; QTRY(dev_tape_run(&tape_, 1)); }
    } else if (rc == 0) {
      const int rc2 = update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_);
      dev_graph_t g = nullptr;
      const int rc3 = dev_graph_end(&g);
      if (rc2 || rc3) { graph_ok_ = false; if (g) dev_graph_destroy(g); QTRY(update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_)); }
      else { graph_ = g; QTRY(dev_graph_launch(graph_)); }
    } else {
      graph_ok_ = false;   // backend cannot capture
      QTRY(update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_));
    }
  } else {
    QTRY(update_amps(out, out + (int64_t)o_ * v_));
    ++eager_iters_;
  }
  return 0;
}

// Lock-step sweeps: record the tape BEFORE the first iteration, on the fragment's own thread and stream (so that the fragments do this side
// by side).  One eager update_amps into the staging buffer settles the workspaces -- it reads the amplitudes and writes only scratch, the
// solver's state is what it was -- and the capture that follows records the launch sequence without executing it.  Every iteration of the
// solve, the first included, then runs from the tape.  A fragment that is not in the replay regime (large, or graphs disabled) is left alone.
int CcsdSolver::prepare_tape(int peers) {
  static const bool peer_hint = [] { const char* e = std::getenv("QEMB_LOCKSTEP_PEERS"); return e && e[0] != '0'; }();      // off: measured slower (below)
  struct Peers { bool on; ~Peers() { if (on) dev_gemm_set_peers(1); } } guard{peer_hint && peers > 1};
  if (guard.on) dev_gemm_set_peers(peers);
  const bool use_diis = false;
  (void)use_diis;
  bool deferred = false;
  if (tape_ || !graph_ok_) return 0;
  // same predicate as iterate_update
  const bool small = (int64_t)o_ * o_ * v_ * v_ <= (int64_t)1 << 22;
  static const bool graphs_enabled = [] {
    const char* e = std::getenv("QEMB_GRAPH");
    if (e) return e[0] != '0';
    const char* pre = std::getenv("LD_PRELOAD");
    if (std::getenv("ROCP_TOOL_LIBRARIES") || (pre && std::strstr(pre, "rocprofiler"))) return false;
    return true;
  }();
  if (!(graphs_enabled && small)) return 0;
  (void)deferred;
  // Record straight away: a capture executes nothing, it only must not allocate -- and in the small-fragment regime update_amps works in buffers that setup() sized
  // (slabs included).  Should a workspace have to grow after all, the capture fails on that allocation: then one eager pass into the staging buffer settles the
  // workspaces -- it reads the amplitudes and writes only scratch -- and the recording is repeated.  (The unconditional eager pass cost every fragment of every sweep
  // ~46 launches.)
  static const bool dry_first = std::getenv("QEMB_TAPE_DRY_PASS") != nullptr;
  for (int attempt = dry_first ? 1 : 0; attempt < 2; ++attempt) {
    if (attempt == 1) { QTRY(update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_)); ++eager_iters_; }
    const int rc = dev_graph_begin(1);
    if (rc != 0) { graph_ok_ = false; return rc < 0 ? rc : 0; }
    const int rc2 = update_amps(ampn_.p, ampn_.p + (int64_t)o_ * v_);
    dev_tape_t t = nullptr;
    const int rc3 = dev_tape_end(&t);
    if (!rc2 && !rc3) { tape_ = t; return 0; }
    if (t) dev_tape_destroy(t);
    if (attempt == 1) { graph_ok_ = false; if (rc2) return rc2; return (rc3 < 0) ? rc3 : 0; }      // (a failed first attempt: whatever went wrong shows again in the eager pass)
  }
  return 0;
}

// DIIS and energy of the iteration whose update iterate_update issued (or that ran as part of a grouped tape run), in three steps with a
// wait of this context's stream between them (iterate_post); the lock-step sweep runs each step for all fragments before the next, so the
// fragments' streams work side by side and the host waits once per step and fragment instead of serialising whole post phases.
int CcsdSolver::iterate_post(double* e_corr, double* normt) {
  QTRY(post_issue());
  QTRY(post_wait(1));
  QTRY(post_extrapolate(normt));
  QTRY(post_wait(2));
  return post_energy(e_corr);
}
// the wait after step 1 / 2: on the host word the fused launch publishes behind its results, else on the stream
int CcsdSolver::post_wait(int step) {
  if (!fused_post()) return dev_sync();
  if (step == 1 && last_use_diis_) return diis_[0].wait_row();
  QTRY(dev_wait_flag(step == 1 ? host_scal_ + 3 : host_scal_ + 2, step == 1 ? seq_push_ : seq_energy_));
  post_pending_ = false;
  return 0;
}
// (the fused launches take up to eight stored vectors and o^2 <= 16384 tiles; QEMB_POST_FUSED=0: the pass-by-pass form, for A/B runs)
bool CcsdSolver::fused_post() const {
  static const bool on = [] { const char* e = std::getenv("QEMB_POST_FUSED"); return !e || e[0] != '0'; }();
  return on && !(last_use_diis_ && diis_[0].space() > 8) && (int64_t)o_ * o_ <= 16384 && v_ <= 4096;
}
// step 1: diff = t_new - t (also the DIIS error vector: trial minus previously returned vector) and its Gram row -- one launch whose last
// workgroup leaves the row (|t_new - t|^2 is its diagonal element) in pinned host memory
int CcsdSolver::post_issue() {
  const int64_t na = n_amp();
  double* out = last_out_;
  if (!host_scal_) { void* q = nullptr; QTRY(dev_pinned_alloc(&q, 4 * sizeof(double))); host_scal_ = (double*)q; for (int k = 0; k < 4; ++k) host_scal_[k] = 0.0; seq_push_ = seq_energy_ = 0; }
  if (!fused_post()) {      // the general path, pass by pass
    double* err = last_use_diis_ ? diis_[0].next_e() : diff_.p;
    QTRY(lincomb2(na, 1.0, out, -1.0, amp_, err));
    if (last_use_diis_) {
      if (out != diis_[0].next_x()) QTRY(dcopy(na, out, diis_[0].next_x()));
      return diis_[0].gram_issue();
    }
    QTRY(dev_dot(na, err, err, scal_.p + 1));
    QTRY(dcopy(na, out, amp_));
    return dev_d2h_async(host_scal_ + 1, scal_.p + 1, sizeof(double));
  }
  if (last_use_diis_) return diis_[0].push_diff_issue(out, amp_);
  const double* self[1] = {diff_.p};
  post_pending_ = true;
  return dev_diis_push(na, out, amp_, diff_, amp_, 1, self, 0, scal_.p + 1, host_scal_ + 1, host_scal_ + 3, ++seq_push_);      // amp = t_new on the way
}
// step 2 (after a wait): the DIIS solve on the host; the extrapolated amplitudes, their tau and the energy
//   E = sum (2 ovov[iajb] - ovov[ibja]) tau[ijab] = <Loovv, tau>   (f_ov = 0)
// in one launch (energy to pinned host memory)
int CcsdSolver::post_extrapolate(double* normt) {
  double nn = 0.0;
  first_ = false;
  if (!fused_post()) {      // beyond what the fused launch takes: the general path
    if (last_use_diis_) QTRY(diis_[0].gram_finish(amp_, false, &nn));
    else nn = host_scal_[1];
    *normt = std::sqrt(nn);
    QTRY(make_tau(t1(), t2(), tau_));
    QTRY(dev_dot((int64_t)o_ * o_ * v_ * v_, Loovv_, tau_, scal_));
    return dev_d2h_async(host_scal_, scal_, sizeof(double));
  }
  double c[8] = {1.0};
  const double* xs[8] = {amp_.p};
  int m = 1;
  if (last_use_diis_) QTRY(diis_[0].coefficients(&m, c, xs, &nn));
  else nn = host_scal_[1];
  *normt = std::sqrt(nn);
  post_pending_ = true;
  return dev_ccsd_extrapolate_energy(o_, v_, m, c, xs, amp_, Loovv_, tau_, scal_, host_scal_, host_scal_ + 2, ++seq_energy_);
}
// step 3 (after a wait)
int CcsdSolver::post_energy(double* e_corr) {
  ecc_ = host_scal_[0];
  *e_corr = ecc_;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Several fragments in LOCK STEP (the small-fragment regime: octane BE2 has six fragments of ~40 orbitals whose iterations are ~110
// dependent launches of 4-5 us each, whatever the fragment count).  Control flow of CcsdSolver::kernel for every solver at once; per
// iteration the update_amps launch sequences of all still-iterating fragments run as ONE merged sequence (dev_tape_run: the same kernel of
// several fragments = one grouped launch), then each fragment does its own DIIS step and energy.  Every fragment performs exactly the
// operations of its own kernel() in the same order -- results are bit-identical to the one-by-one sweep.
// ctx[f]: the execution context (stream + workspaces) fragment f was prepared on and keeps for its own steps; the grouped launches run
// on the calling thread's context, which is restored on return.
int ccsd_kernel_lockstep(const std::vector<CcsdSolver*>& s, const std::vector<CcsdOptions>& opt, const std::vector<int>& ctx, int home_ctx,
                         std::vector<double>& e_corr, std::vector<int>& n_iter, std::vector<char>& converged, LockstepStats* stats) {
  const int F = (int)s.size();
  e_corr.assign(F, 0.0); n_iter.assign(F, 0); converged.assign(F, 0);
  std::vector<double> eold(F), e(F), normt(F, 0.0);
  std::vector<char> active(F, 1);
  struct Rebind { int home; ~Rebind() { (void)dev_ctx_bind(home); } } rebind{home_ctx};
  for (int f = 0; f < F; ++f) {
    QTRY(dev_ctx_bind(ctx[f]));
    s[f]->diis_.clear();
    if (opt[f].diis_space > 1) { s[f]->diis_.emplace_back(opt[f].diis_space, s[f]->n_amp()); QTRY(s[f]->diis_[0].init()); }
    eold[f] = e[f] = s[f]->ecc_;
  }
  int max_cycle = 0;
  for (int f = 0; f < F; ++f) max_cycle = std::max(max_cycle, opt[f].max_cycle);
  for (int it = 1; it <= max_cycle; ++it) {
    std::vector<dev_tape_t> tapes;
    std::vector<int> tape_ctx;
    bool any = false;
    for (int f = 0; f < F; ++f) {
      if (!active[f]) continue;
      any = true;
      QTRY(dev_ctx_bind(ctx[f]));
      bool deferred = false;
      QTRY(s[f]->iterate_update(true, true, &deferred));
      if (deferred) { tapes.push_back(s[f]->tape_); tape_ctx.push_back(ctx[f]); }
    }
    if (!any) break;
    // the fragments' own streams must have finished what the tapes read (the previous post steps end with a host sync; an eager or
    // capturing update above is followed by its own post step below, which syncs too) -- and the merged run must finish before the post steps
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    bool batched_post = false;
    int nactive = 0;
    for (int f = 0; f < F; ++f) nactive += active[f] ? 1 : 0;
    if (!tapes.empty()) {
      // The merged sequence is issued as `nsplit` sequences of tapes.size() / nsplit members each, on the streams of the first nsplit
      // fragments' contexts: grouped launches of six small fragments are bound by workgroup dispatch (8-15 us each), two sequences of three
      // members overlap on the device (QEMB_LOCKSTEP_SPLIT; measured: 13.2 ms per octane sweep in one sequence, 14.8 in two, 16.0 in three -- the default stays one).
      static const int want_split = std::getenv("QEMB_LOCKSTEP_SPLIT") ? std::max(1, std::atoi(std::getenv("QEMB_LOCKSTEP_SPLIT"))) : 1;
      const int nt = (int)tapes.size();
      const int nsplit = (nt >= 4) ? std::min(want_split, nt / 2) : 1;
      std::vector<int> run_ctx;
      for (int k = 0; k < nsplit; ++k) {
        const int a = (int)((long long)nt * k / nsplit), b = (int)((long long)nt * (k + 1) / nsplit);
        const int c = (nsplit == 1) ? home_ctx : tape_ctx[a];
        QTRY(dev_ctx_bind(c));
        QTRY(dev_tape_run(tapes.data() + a, b - a));
        run_ctx.push_back(c);
        if (stats) {
          long long l = 0, g = 0, ops = 0;
          (void)dev_tape_last_stats(&l, &g, &ops);
          stats->launches += l; stats->grouped += g; stats->operations += ops;
          stats->max_group = std::max<long long>(stats->max_group, (long long)(b - a));
        }
      }
      if (stats) stats->merged_runs += 1;
      // Every active fragment ran in ONE merged sequence on the home stream and ends its iteration with the fused launches: those are collected and issued -- grouped,
      // four launches for all fragments -- on the same stream, behind the merged run: no wait in between, and the host words tell when each fragment is through.
      batched_post = (nsplit == 1 && nt == nactive);
      for (int f = 0; f < F && batched_post; ++f) if (active[f] && !s[f]->fused_post()) batched_post = false;
      if (!batched_post) for (int c : run_ctx) { QTRY(dev_ctx_bind(c)); QTRY(dev_sync()); }
    }
    const double t1 = now();
    if (stats) stats->ms_tapes += t1 - t0;
    // QEMB_BATCH_TRACE=2: iterations that take longer than 5 ms are reported with their parts (where did a slow sweep spend its time?)
    static const bool trace_iters = std::getenv("QEMB_BATCH_TRACE") && std::atoi(std::getenv("QEMB_BATCH_TRACE")) >= 2;
    struct IterTrace { bool on; int it; double t0, t1; double (*now)(); ~IterTrace() { const double t = now(); if (on && t - t0 > 5.0) std::fprintf(stderr, "[qemb lockstep] iteration %d took %.2f ms: tape issue %.2f, post steps %.2f\n", it, t - t0, t1 - t0, t - t1); } } iter_trace{trace_iters, it, t0, t1, +now};
    struct PostTime { LockstepStats* st; double t1; double (*now)(); ~PostTime() { if (st) st->ms_post += now() - t1; } } post_time{stats, t1, +now};
    if (batched_post) {
      struct Flush { bool open = false; ~Flush() { if (open) (void)dev_batch_flush(); } } guard;      // (an error exit must not leave the collector open)
      QTRY(dev_ctx_bind(home_ctx));
      QTRY(dev_batch_begin()); guard.open = true;
      for (int f = 0; f < F; ++f) if (active[f]) QTRY(s[f]->post_issue());
      guard.open = false; QTRY(dev_batch_flush());
      QTRY(dev_batch_begin()); guard.open = true;
      for (int f = 0; f < F; ++f) if (active[f]) { QTRY(s[f]->post_wait(1)); QTRY(s[f]->post_extrapolate(&normt[f])); }
      guard.open = false; QTRY(dev_batch_flush());
    } else {
      for (int f = 0; f < F; ++f) if (active[f]) { QTRY(dev_ctx_bind(ctx[f])); QTRY(s[f]->post_issue()); }
      for (int f = 0; f < F; ++f) if (active[f]) { QTRY(dev_ctx_bind(ctx[f])); QTRY(s[f]->post_wait(1)); QTRY(s[f]->post_extrapolate(&normt[f])); }
    }
    for (int f = 0; f < F; ++f) {
      if (!active[f]) continue;
      if (!batched_post) QTRY(dev_ctx_bind(ctx[f]));
      QTRY(s[f]->post_wait(2));
      QTRY(s[f]->post_energy(&e[f]));
      n_iter[f] = it;
      if (opt[f].verbose > 0) std::fprintf(stderr, "[qemb ccsd lockstep] frag %d cycle %3d  E(corr) = %.12f  dE = %.3e  |dt| = %.3e\n", f, it, e[f], e[f] - eold[f], normt[f]);
      if (!std::isfinite(e[f])) { set_error("CCSD diverged (non-finite energy)"); return QEMB_ERR_NUMERIC; }
      if (std::fabs(e[f] - eold[f]) < opt[f].conv_tol && normt[f] < opt[f].conv_tol_normt) { converged[f] = 1; active[f] = 0; }
      else if (it >= opt[f].max_cycle) active[f] = 0;
      eold[f] = e[f];
    }
  }
  for (int f = 0; f < F; ++f) { e_corr[f] = e[f]; s[f]->diis_.clear(); }
  return 0;
}

int CcsdSolver::kernel(const CcsdOptions& opt, double* e_corr, int* n_iter, bool* converged) {
  diis_.clear();
  if (opt.diis_space > 1) {
    diis_.emplace_back(opt.diis_space, n_amp());
    QTRY(diis_[0].init());
  }
  double eold = ecc_, e = ecc_, normt = 0.0;
  *converged = false;
  int it = 0;
  for (it = 1; it <= opt.max_cycle; ++it) {
    QTRY(iterate(&e, &normt));
    if (opt.verbose > 0) std::fprintf(stderr, "[qemb ccsd] cycle %3d  E(corr) = %.12f  dE = %.3e  |dt| = %.3e\n", it, e, e - eold, normt);
    if (!std::isfinite(e)) { set_error("CCSD diverged (non-finite energy)"); return QEMB_ERR_NUMERIC; }
    if (std::fabs(e - eold) < opt.conv_tol && normt < opt.conv_tol_normt) { *converged = true; break; }
    eold = e;
  }
  *e_corr = e;
  *n_iter = it > opt.max_cycle ? opt.max_cycle : it;
  diis_.clear();
  return 0;
}

int CcsdSolver::export_block(const char* name, double* host, int64_t nelem) {
  const std::string nm(name ? name : "");
  const DBuf* b = nullptr;
  if (nm == "oooo") b = &I_.oooo; else if (nm == "ovoo") b = &I_.ovoo; else if (nm == "ovov") b = &I_.ovov;
  else if (nm == "ovvv") b = &I_.ovvv; else if (nm == "Vl") b = &I_.Vl; else if (nm == "Vp") b = &I_.Vp; else if (nm == "Vm") b = &I_.Vm; else if (nm == "W1base") b = &W1base_;
  else if (nm == "W2base") b = &W2base_; else if (nm == "eo") b = &eo_; else if (nm == "ev") b = &ev_;
  if (!b || !b->p) { set_error("export_block: unknown block name"); return QEMB_ERR_ARG; }
  if (nelem != b->n) { set_error("export_block: element count mismatch"); return QEMB_ERR_ARG; }
  return dev_d2h(host, b->p, sizeof(double) * nelem);
}

// Z1[i,P] = sum_{ajb} G[i,a,j,b] (Pa|jb),  Z2[a,P] = sum_{ijb} G[i,a,j,b] (Pi|jb),  G = 2 tau[ijab] - tau[jiab]
// (the contracted form of make_rdm2_urlx(with_dm1=False) + the 'ijkl,pi,qj,rk,sl' rotation of helper.py:307)
int CcsdSolver::energy_intermediates(std::vector<double>& Z1, std::vector<double>& Z2) {
  const int64_t o = o_, v = v_, nf = nf_, N2 = o * o * v * v;
  Z1.assign((size_t)(o * nf), 0.0); Z2.assign((size_t)(v * nf), 0.0);
  if (nf <= 0) return 0;
  QTRY(make_tau(t1(), t2(), tau_));
  QTRY(perm4(S_, tau_, o, o, v, v, 0, 2, 1, 3, 2.0, 0.0));     // 2 tau[ijab] at [i,a,j,b]
  QTRY(perm4(S_, tau_, o, o, v, v, 1, 2, 0, 3, -1.0, 1.0));    // - tau[jiab]: out[i,a,j,b] = in[j,i,a,b]
  DBuf z1, z2, a2p;
  QTRY(z1.alloc(o * nf)); QTRY(z2.alloc(v * nf)); QTRY(a2p.alloc(o * o * v * nf));
  QTRY(gemm_nn(o, nf, v * o * v, 1.0, S_, I_.A1, 0.0, z1));    // Z1[i,P] = G[i,(ajb)] A1[(ajb),P]
  // Z2[a,P] = sum_{(jbi)} G[(jbi),a] A2p[(jbi),P],  A2p[j,b,i,P] = A2[i,j,b,P]
  QTRY(perm4(a2p, I_.A2, o, o, v, nf, 1, 2, 0, 3));
  QTRY(gemm_tn(v, nf, o * v * o, 1.0, S_, a2p, 0.0, z2));
  QTRY(dev_d2h(Z1.data(), z1, sizeof(double) * o * nf));
  QTRY(dev_d2h(Z2.data(), z2, sizeof(double) * v * nf));
  (void)N2;
  return 0;
}

}  // namespace qemb
