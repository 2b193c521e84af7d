/* qemb_hip_ops.h -- device-pointer primitives, device timers and measurement / tuning hooks of libqemb_hip.so.
 *
 * NOT part of the product ABI (include/qemb_hip.h): these entry points exist so that the parity tests can compare every kernel the
 * drivers are composed of with NumPy (tests/test_gpu_ops.py), and so that bench.py / tools/ can time single kernels.  All pointers
 * are DEVICE pointers (qemb_malloc) unless stated.  The tuning setters (qemb_set_gemm_*) act on the CALLING HOST THREAD only -- a host
 * thread drives one execution context (HIP stream), so a setter can never change the GEMMs in flight on another stream.
 */
#ifndef QEMB_HIP_OPS_H
#define QEMB_HIP_OPS_H
#include "qemb_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* device-time laps measured with HIP events on the library stream (slot ids: see QEMB_TIMER_*) */
#define QEMB_TIMER_LADDER 0
#define QEMB_TIMER_RINGS 1
#define QEMB_TIMER_ITER 2
#define QEMB_TIMER_AO2MO 3
#define QEMB_TIMER_SCF 4
#define QEMB_TIMER_GEMM_ANY 5
#define QEMB_TIMER_SCHMIDT 6
#define QEMB_TIMER_DF 7
int qemb_timer_begin(int slot);
int qemb_timer_end(int slot);
int qemb_timer_read(int slot, double* total_ms, int64_t* count);
int qemb_timer_reset(int slot);
int qemb_alloc_stats(long long* n_driver_allocs, long long* n_driver_frees, double* ms_in_driver_calls, double* gb_allocated, int reset);   /* pool misses (real hipMalloc calls) and real hipFree calls since the last reset */
int qemb_gemm_flop_count(double* flops, int reset);   /* 2 M N K batch summed over every FP64 MFMA product issued since the last reset (executed flops of a region; measurement hook) */
int qemb_tape_cache_counters(int64_t* reused, int64_t* recorded, int reset);   /* lock-step sweeps: recorded amplitude updates kept from the last solve of a fragment / recorded anew, since the last reset (measurement hook) */
int qemb_ctx_timer_read(int ctx, int slot, double* total_ms, int64_t* count, int reset);   /* timers of an idle execution context */
int qemb_timer_live_events(int slot);      /* event pairs held by the calling context's slot (bounded by recycling)   */

/* ---------------------------------------------------------------- device-pointer primitives ---- */
/* (the kernels the drivers below are composed of; exported so the parity tests can hit each one)    */

/* C[b] = alpha*op(A[b])*op(B[b]) + beta*C[b] on v_mfma_f64_16x16x4_f64.
 * a_kcontig: A(m,k)=A[m*lda+k] else A[k*lda+m];  b_kcontig: B(k,n)=B[n*ldb+k] else B[k*ldb+n].      */
int qemb_op_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int a_kcontig,
                 int64_t strideA, const double* B, int64_t ldb, int b_kcontig, int64_t strideB, double beta,
                 double* C, int64_t ldc, int64_t strideC, int64_t batch);
/* C = A B with the rows of A (stored K x M, i.e. !a_kcontig) read through the slab-aware loader: A(m,k) = A[k*lda + m + (m / a_slab) * a_slab_skip].  With
 * lda = a_slab = n and a_slab_skip = n^2 - n the rows (pair, q) of a stack of n x n slabs X[pair][k][q] form ONE tall operand: the last quarter transform
 * of the embedding -> MO transformation (C^T . slab for every pair, csrc/ccsd.cpp mo_transform) as a flat product on the tall tile.  cfg as qemb_set_gemm_config. */
int qemb_op_gemm_slab_rows(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, int64_t a_slab, int64_t a_slab_skip, const double* B, int64_t ldb,
                           int b_kcontig, double* C, int64_t ldc, int cfg);
/* -1 = automatic tile choice; 0..99 force a tile configuration (calling host thread only); 2xx = the round-1 main loop of the same tiles
 * (A/B measurements); 3xx = diagnostic instantiations that stamp s_memtime (qemb_op_gemm_stamps); 4xx / 5xx / 6xx = ABLATION instantiations
 * that leave memory traffic out and return WRONG products by construction (tools/gemm_ablation.py) -- measurement aids, reachable through
 * this hook only, never selected by the dispatcher or by any driver.                                                                      */
int qemb_set_gemm_config(int cfg);
/* calibration: sustained v_mfma_f64_16x16x4_f64 rate of the chip, registers only (TFLOP/s)           */
int qemb_mfma_f64_peak(int iters, int blocks_per_cu, double* tflops);
int qemb_set_gemm_splitk(int enabled);    /* automatic split-K for few-tile / long-K products (default on) */
/* One product (device pointers) timed on its own, with the sustained shader clock of the launch: every workgroup records its
 * s_memtime ticks, clock_ghz = sum(ticks) / (256 CUs x time) -- the clock itself when one workgroup is resident per CU (tile configs
 * 13/15/23/25), a multiple of it otherwise.  Synchronises; a measuring aid for bench.py / tools, not part of the solver path.        */
int qemb_op_gemm_probe(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, int a_kcontig, const double* B, int64_t ldb, int b_kcontig,
                       double* C, int64_t ldc, int cfg, int ksplit, double* ms, double* clock_ghz, int64_t* workgroups);
/* diagnostic tile configurations 313 / 315 / 304 (8-wave tiles, both operands K-contiguous): per-wave s_memtime sums of one launch, averaged over its
 * waves: out7 = [cycles in k-steps 0..2 of the tiles, from there to past the per-tile barrier, in the last k-step, kernel ms, wave entry -> end of the
 * main loop, wave entry -> exit, workgroups] (sums over the k-tiles of a wave) */
int qemb_op_gemm_stamps(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int cfg,
                        int ksplit, double* out7);
/* the tile configuration and split-K factor the CCSD driver picks for a product of `rows` packed pair rows by `cols` columns (pp-ladder,
 * tau-side dressing): introspection for tests and tools, no device call */
int qemb_pair_gemm_choice(int64_t rows, int64_t cols, int* cfg, int* ksplit);
int qemb_set_gemm_ksplit(int ksplit);      /* explicit split-K factor for qemb_op_gemm (0 = automatic) */
/* out[sum ik*so[k]] = alpha*in[sum ik*si[k]] + beta*out[...], 0<=ik<dim[k], 4 dims                  */
int qemb_op_copy4(const int64_t dim[4], const double* in, const int64_t si[4], double* out,
                  const int64_t so[4], double alpha, double beta);
int qemb_op_outer4(const int64_t dim[4], const double* u, int64_t su0, int64_t su2, const double* v,
                   int64_t sv1, int64_t sv3, double* out, const int64_t so[4], double alpha, double beta);
int qemb_op_div_denom(double* x, int64_t d0, int64_t d1, int64_t d2, int64_t d3, const double* ea,
                      const double* eb, const double* ec, const double* ed);
/* (+/-) pair-packed pp-ladder helpers, P(x,y) = x(x+1)/2+y (x>=y), Q(x,y) = x(x-1)/2+y (x>y):
 * Vp[P(ab),P(cd)] = (ac|bd)+(ad|bc), Vm[Q(ab),Q(cd)] = (ac|bd)-(ad|bc) from the n^4 MO tensor (virtuals offset o);
 * Tp[P(ij),P(cd)] = w(tau_ijcd+tau_ijdc), w = 1/2 | 1/4 (c==d), Tm[Q(ij),Q(cd)] = (tau_ijcd-tau_ijdc)/2;
 * scatter: t2[ijab] += Rp+Rm, t2[ijba] += Rp-Rm, t2[jiab] += Rp-Rm, t2[jiba] += Rp+Rm                         */
int qemb_op_ladder_pack_vvvv(int64_t n, int64_t o, const double* M, double* Vp, int64_t ldp, double* Vm, int64_t ldm);
int qemb_op_ladder_pack_tau(int64_t o, int64_t v, const double* tau, double* Tp, int64_t ldp, double* Tm, int64_t ldm);
int qemb_op_ladder_scatter_pm(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, double* t2);
int qemb_op_dot(int64_t n, const double* x, const double* y, double* out_dev);
int qemb_op_absmax(int64_t n, const double* x, double* out_dev);
int qemb_op_gemv_rows(int64_t rows, int64_t cols, const double* T, int64_t ldt, const double* x, double* y,
                      double alpha, double beta);
int qemb_op_gemv_rows_batched(int64_t rows, int64_t cols, int64_t nbatch, const double* T, int64_t ldt, int64_t strideT,
                              const double* x, int64_t stridex, double* y, double alpha, double beta);
int qemb_op_contract_mid(int64_t outer, int64_t mid, int64_t inner, const double* T, const double* x,
                         double* Y, int64_t ldy, double alpha, double beta);
int qemb_op_unpack_s4(int64_t n, const double* s4, double* s1);
int qemb_op_pack_s4(int64_t n, const double* s1, double* s4);
int qemb_op_unpack_s8_to_s4(int64_t n, const double* s8, double* s4);
/* A[r][c] = A[c][r], r < c (completes a SYRK-style result computed on and below the diagonal) */
int qemb_op_mirror_lower(int64_t n, double* A, int64_t lda);
/* exchange matrix K[p,r] = sum (pq|rs) D[q,s] from the half-unpacked tensor H[P(p,q)][r][s] (scf.hf.dot_eri_dm's K at helper.py:64) */
int qemb_op_k_from_pairs(int64_t n, const double* H, const double* D, double* K);
/* the same K and the packed Coulomb vector Jp[P(p,q)] = sum_{r>=s} (pq|rs) Dp[P(r,s)] in ONE pass over the 4-fold packed block S4
 * (J and K of scf.hf.dot_eri_dm, helper.py:64); Dp = D + D^T off the diagonal, D on it, packed; Dp / Jp may be null; n <= 1024 */
int qemb_op_jk_from_packed(int64_t n, const double* S4, const double* D, const double* Dp, double* Jp, double* K);
/* (+/-) pair packing of the last two indices of in[rows][v][v] (Op: c >= d sums, Om: c > d differences; rows padded to ldp / ldm)
 * and the inverse scatter of packed pair ROWS: out[i,j,:] = Xp + Xm, out[j,i,:] = Xp - Xm */
int qemb_op_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm);
int qemb_op_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out);
/* hole-hole ladder through packed pairs: the (+/-) packed images of W[k,l,i,j] (Ap[P(ij)][P(kl)] = W[klij] + W[klji], W[kkij] on k = l;
 * Am[Q(ij)][Q(kl)] = W[klij] - W[klji]) and the scatter of TWO pairs of packed result rows, p = Rp + f Hp (f = 2 on a = b), m = Rm + Hm,
 * assigned to (assign != 0) or accumulated into t2 */
int qemb_op_pack_w_pm(int64_t o, const double* W, double* Ap, int64_t lda_p, double* Am, int64_t lda_m);
int qemb_op_ladder_scatter_pm2(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, const double* Hp, const double* Hm,
                               int assign, double* t2);
int qemb_op_lincomb2(int64_t n, double a, const double* x, double b, const double* y, double beta, double* out);   /* out = a x + b y + beta out */
/* Single-pass kernels of the CCSD amplitude update (csrc/ccsd.cpp), device pointers:
 *   small_k_update: C[z][m][n] += alpha sum_k A[z][k][m] B[z][k][n]  (K = n_occ; batch strides sA / sB / sC, 0 shares an operand)
 *   ccsd_ph_layouts: from t2[o][o][v][v] and t1 in one pass T[k,c,j,b] = t2[k,j,c,b], Tp = t2[k,j,b,c], S = 2T - Tp,
 *                    Ut = S - 2 t1[j,c] t1[k,b], Tpt = Tp + 2 t1[j,c] t1[k,b] (all [o][v][o][v]) and Th[k,j,c,b] = 2 t2[k,j,b,c] - t2[k,j,c,b]
 *   ccsd_y_traces:  Y[a,c] = 2 sum_k ZC[k,k,a,c] - sum_k ZB[k,c,a,k]   (ZC [o][o][v][v], ZB [o][v][v][o])
 * and of the semi-sparse DF transform: gather_rows dst[r,:] = idx[r] >= 0 ? src[idx[r],:] : 0 (idx: int64 on the device), scale_rows x[r,:] *= s[r] */
int qemb_op_small_k_update(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sA, const double* B, int64_t sB, double* C, int64_t sC);
int qemb_op_ccsd_ph_layouts(int64_t o, int64_t v, const double* t2, const double* t1, double* T, double* Tp, double* S, double* Ut, double* Tpt, double* Th);
int qemb_op_ccsd_y_traces(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y);
/* Fused passes of the CCSD amplitude update (csrc/ccsd.cpp update_amps), device pointers:
 *   copy4_two: qemb_op_copy4 with base (NULL: out) and a second output of the same pass, addressed like out: out2 = c2a in2 + c2b (value written to out)
 *   scatter_pm_rows_add / ccsd_y_traces_add: the plain ops with an addend laid out like the result
 *   pack_w_pm_sum: pack_w_pm of W[k,l,i,j] = Wt[i,j,k,l] + X[i,j,k,l] + At[j,i,k,l] + At[i,j,l,k], W never stored
 *   ccsd_t1_small: t1n[i,a] = sum_c t1[i,c] Lvv[a,c] - sum_k Loo[k,i] t1[k,a] + sum_k (sum_c t1[i,c] Fov[k,c]) t1[k,a]
 *   gemv_rows2: y = alpha (T1 x1 + T2 x2) + beta y
 *   ccsd_finish_t2_rings: F[ijab] = U[ijab] + RS[iajb] - M[iajb] / 2 - M[ibja];  t2n[ijab] = t2n[jiba] = (t2n[ijab] + OV[ijab] + F[ijab] + F[jiba]) / D (i >= j);
 *                         t1n[ia] /= eo[i] - ev[a] (t1n may be NULL) */
int qemb_op_copy4_two(const int64_t dim[4], const double* in, const int64_t si[4], double* out, const int64_t so[4], double alpha, double beta, const double* base,
                      double* out2, const double* in2, double c2a, double c2b);
int qemb_op_scatter_pm_rows_add(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out, const double* add);
int qemb_op_ccsd_y_traces_add(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add);
int qemb_op_pack_w_pm_sum(int64_t o, const double* Wt, const double* X, const double* At, double* Ap, int64_t lda_p, double* Am, int64_t lda_m);
int qemb_op_ccsd_t1_small(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, double* t1n);
/*   ccsd_t1_assemble: t1n = [ccsd_t1_small] + S[(ia),:] . Fov + Lph1[(ia),:] . t1 + sum_s PA[s] - sum_s PB[s]  (PA / PB: SA / SB slabs of o v doubles, strideA / strideB apart)
 *   gemv_rows_two: two independent matrix-vector passes in one launch;  ccsd_y_traces_slabs: Y = traces + scale * sum of S slabs of `add` */
int qemb_op_ccsd_t1_assemble(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, const double* S, const double* Lph1,
                             const double* PA, int SA, int64_t strideA, const double* PB, int SB, int64_t strideB, double* t1n);
int qemb_op_gemv_rows_two(int64_t rows1, int64_t cols1, const double* T1, int64_t ld1, const double* x1, double* y1, double a1, double b1,
                          int64_t rows2, int64_t cols2, const double* T2, int64_t ld2, const double* x2, double* y2, double a2, double b2);
int qemb_op_ccsd_y_traces_slabs(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add, int S, int64_t stride, double scale);
int qemb_op_gemv_rows2(int64_t rows, int64_t cols, const double* T1, int64_t ld1, const double* x1, const double* T2, int64_t ld2, const double* x2, double* y, double alpha, double beta);
int qemb_op_ccsd_finish_t2_rings(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* RS, const double* M, const double* eo, const double* ev, double* t1n);
/* The two single-launch ends of a CCSD iteration (csrc/ccsd.cpp post_issue / post_extrapolate), device pointers except where noted:
 *   diis_push: e = trial - prev, xcopy = trial (xcopy may be NULL), row[j] = <e, ys[j]> for j < m <= 8 with ys[self] == e; `ys` is a HOST array of
 *              m device pointers; the row goes to row_dev (device, m doubles) and row_host (HOST, m doubles: the wrapper waits for it)
 *   ccsd_extrapolate_energy: amp = sum_k coef[k] xs[k] over [t1 | t2] (coef: HOST, xs: HOST array of device pointers, nterms <= 8), tau = t2 + t1 (x) t1,
 *              *e_host = <L, tau> (HOST double; the wrapper waits) */
int qemb_op_diis_push(int64_t n, const double* trial, const double* prev, double* e, double* xcopy, int m, const double* const* ys, int self, double* row_dev, double* row_host);
int qemb_op_ccsd_extrapolate_energy(int64_t o, int64_t v, int nterms, const double* coef, const double* const* xs, double* amp, const double* L, double* tau, double* e_host);
int qemb_op_gather_rows(int64_t nrows, int64_t len, const int64_t* idx_dev, const double* src, int64_t ld, double* dst);
int qemb_op_scale_rows(int64_t nrows, int64_t len, double* x, const double* s);
/* pair-packed MO transformation helpers (half the flops of the four-index ao2mo.kernel call of PySCF's cc.ao2mo(), which
 * solve_ccsd reaches at molbe/solver.py:900): row gather x >= y; the same fused with the unpack of the pair column; block gathers
 * from the pair-first MO tensor Mp[P(p,q)][r][s] and from the 3/4-transformed tensor T[P(r,s)][c][x]; (+/-) ladder operands. */
int qemb_op_pack_pair_rows(int64_t n, int64_t ncols, const double* in, double* out);
int qemb_op_unpack_tril_pair_rows(int64_t nr, int64_t n, const double* in, double* full);
int qemb_op_extract_pf(int64_t n, const double* Mp, int64_t p0, int64_t q0, int64_t r0, int64_t s0, int64_t sp, int64_t sq,
                       int64_t sr, int64_t ss, double* out);
int qemb_op_extract_pf_t(int64_t n, const double* T, int64_t x0, int64_t r0, int64_t s0, int64_t c0, int64_t sx, int64_t sr,
                         int64_t ss, int64_t sc, double* out);
int qemb_op_ladder_pack_vvvv_pf(int64_t n, int64_t o, const double* Mp, double* Vp, int64_t ldp, double* Vm, int64_t ldm);
int qemb_op_unpack_tril_rows(int64_t rows, int64_t n, const double* packed, double* full);
int qemb_op_pack_tril_rows(int64_t rows, int64_t n, const double* full, double* packed);
int qemb_op_jacobi_eigh(int64_t n, double* A, double* w, double* V, int* sweeps);
/* The SCF cycle of a SMALL fragment in two fused launches (round 5; n <= qemb_op_scf_fused_max(), 80 on the device; reference: molbe/helper.py:73-151, PySCF scf.hf.kernel):
 * qemb_op_scf_fock_small: F = h + J - K/2, err = F D - D F, scal2 = [sum (h + F) o D, sum err^2] (device);
 * qemb_op_jacobi_eigh_in_basis: eigenproblem of F in the orthonormal basis Cp (NULL: as given) -- w ascending, C_out = Cp V, the same to C2_out (NULL or any buffer, Cp itself
 * allowed), dm_out (NULL or) 2 C_occ C_occ^T of the lowest nocc columns;  qemb_op_pack_density_sym: Dp[P(r,s)] = D[r,s] + D[s,r] (r > s), D[r,r]. */
int qemb_op_scf_fused_max(void);
int qemb_op_jacobi_eigh_in_basis(int64_t n, const double* F, const double* Cp, double* w, double* C_out, double* C2_out, int nocc, double* dm_out, double stop_below, int* sweeps);
int qemb_op_scf_fock_small(int64_t n, const double* h, const double* J, const double* K, const double* D, double* F, double* err, double* scal2);
int qemb_op_pack_density_sym(int64_t n, const double* D, double* Dp);
int qemb_op_jacobi_svd(int64_t m, int64_t n, double* G, double* s, double* U, double* V, int* sweeps);
int qemb_op_cholesky_lower(int64_t n, double* A);
int qemb_op_tri_inverse_lower(int64_t n, const double* L, double* Linv);


/* measurement hooks: set up SCF + integrals once, then run/timed single CCSD iterations                */
int qemb_frag_prepare_ccsd(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts);
int qemb_frag_ccsd_iterate(qemb_frag_t f, int niter, double* e_corr, double* normt);
int qemb_frag_ccsd_reset(qemb_frag_t f);
/* copy one MO-integral block of the prepared CCSD problem to the host: "oooo" "ovoo" "ovov" "ovvv" "Vl" (= (ac|bd) at
 * [a,b,c,d]) "W1base" (= ovvo[k,c,a,i] at [i,a,k,c]) "W2base" (= oovv[k,i,a,c] at [i,a,k,c]) "eo" "ev" "Vp" "Vm" ((+/-) pair-packed ladder operands) "mo_coeff" (n x n)                     */
int qemb_frag_ccsd_export(qemb_frag_t f, const char* name, double* host, int64_t nelem);


#ifdef __cplusplus
}
#endif
#endif /* QEMB_HIP_OPS_H */
