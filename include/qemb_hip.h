/* qemb_hip.h -- C ABI of libqemb_hip.so: the MI355X (gfx950) per-fragment embedding solver that sits behind
 * QuEmb's Frags / solver / int_transform seams (troyvvgroup/quemb).
 *
 * Conventions (reference precedent: shared/external/unrestricted_utils.py:142-160 -- caller-allocated
 * numpy buffers handed to a C function as plain pointers):
 *   - plain C types only; FP64; row-major; every HOST buffer (in and out) is allocated by the caller;
 *   - device-resident state lives behind opaque handles with explicit *_free;
 *   - every function returns 0 on success and <0 on failure; qemb_last_error() returns the message
 *     (the reference's natives throw C++ exceptions mapped to Python -- _cpp/eri_sparse_DF.cpp:40-62 --
 *     a C ABI returns a status instead and the Python shim raises);
 *   - pair index ij = i(i+1)/2 + j, i >= j (shared/helper.py:260-276, _cpp/indexers.hpp:75-79);
 *   - embedding orbitals are ordered fragment sites first, then bath (molbe/pfrag.py:489-491).
 *   - There is NO CPU fallback: without a visible HIP device qemb_init() fails.
 */
#ifndef QEMB_HIP_H
#define QEMB_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define QEMB_OK 0
#define QEMB_ERR_ARG (-1)
#define QEMB_ERR_ALLOC (-2)
#define QEMB_ERR_DEVICE (-3)
#define QEMB_ERR_NOCONV (-4)
#define QEMB_ERR_NUMERIC (-5)

/* ---------------------------------------------------------------- library / device ------------- */
int qemb_init(int device);                 /* select the GPU, create the library stream            */
const char* qemb_last_error(void);
const char* qemb_backend(void);            /* "hip-gfx950"                                         */
int qemb_sync(void);
int qemb_mem_info(size_t* free_bytes, size_t* total_bytes);

/* raw device buffers (for callers that keep tensors resident, e.g. bench.py / multi-fragment sweeps) */
int qemb_malloc(void** dptr, size_t bytes);
int qemb_free(void* dptr);                 /* parks the block for reuse (caching allocator)       */
int qemb_trim(void);                       /* hand every parked block back to the driver          */
int qemb_h2d(void* dptr, const void* host, size_t bytes);
int qemb_d2h(void* host, const void* dptr, size_t bytes);
int qemb_d2d(void* dst, const void* src, size_t bytes);

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

/* ---------------------------------------------------------------- device-pointer primitives ---- */
/* (the kernels the drivers below are composed of; exported so the parity tests can hit each one)    */

/* C[b] = alpha*op(A[b])*op(B[b]) + beta*C[b] on v_mfma_f64_16x16x4_f64.
 * a_kcontig: A(m,k)=A[m*lda+k] else A[k*lda+m];  b_kcontig: B(k,n)=B[n*ldb+k] else B[k*ldb+n].      */
int qemb_op_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int a_kcontig,
                 int64_t strideA, const double* B, int64_t ldb, int b_kcontig, int64_t strideB, double beta,
                 double* C, int64_t ldc, int64_t strideC, int64_t batch);
int qemb_set_gemm_config(int cfg);         /* -1 = automatic tile choice; >=0 forces a tile config  */
/* calibration: sustained v_mfma_f64_16x16x4_f64 rate of the chip, registers only (TFLOP/s)           */
int qemb_mfma_f64_peak(int iters, int blocks_per_cu, double* tflops);
int qemb_set_gemm_splitk(int enabled);
/* One product (device pointers) timed on its own, with the sustained shader clock of the launch: every workgroup records its
 * s_memtime ticks, clock_ghz = sum(ticks) / (256 CUs x time) -- the clock itself when one workgroup is resident per CU (tile configs
 * 13/15/23/25), a multiple of it otherwise.  Synchronises; a measuring aid for bench.py / tools, not part of the solver path.        */
int qemb_op_gemm_probe(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, int a_kcontig, const double* B, int64_t ldb, int b_kcontig,
                       double* C, int64_t ldc, int cfg, int ksplit, double* ms, double* clock_ghz, int64_t* workgroups);
int qemb_set_gemm_ksplit(int ksplit);      /* explicit split-K factor for qemb_op_gemm (0 = automatic) */     /* split-K for few-tile / long-K products (default on)   */
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
/* (+/-) pair packing of the last two indices of in[rows][v][v] (Op: c >= d sums, Om: c > d differences; rows padded to ldp / ldm)
 * and the inverse scatter of packed pair ROWS: out[i,j,:] = Xp + Xm, out[j,i,:] = Xp - Xm */
int qemb_op_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm);
int qemb_op_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out);
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
int qemb_op_gather_rows(int64_t nrows, int64_t len, const int64_t* idx_dev, const double* src, int64_t ld, double* dst);
int qemb_op_scale_rows(int64_t nrows, int64_t len, double* x, const double* s);
/* Execution contexts (one HIP stream + workspaces + block cache each; no reference counterpart -- the reference overlaps
 * fragments with a process pool, be_parallel.py:484).  qemb_ctx_count(n) makes contexts 0..n-1 available (0 = default) and
 * returns how many exist (or < 0); qemb_ctx_bind(k) binds the CALLING host thread to context k, so that several host
 * threads can each drive a fragment on their own stream; qemb_ctx_timer_read reads the device timers of an idle context. */
int qemb_ctx_count(int n);
int qemb_ctx_bind(int k);
int qemb_ctx_timer_read(int ctx, int slot, double* total_ms, int64_t* count, int reset);
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
int qemb_op_jacobi_svd(int64_t m, int64_t n, double* G, double* s, double* U, double* V, int* sweeps);
int qemb_op_cholesky_lower(int64_t n, double* A);
int qemb_op_tri_inverse_lower(int64_t n, const double* L, double* Linv);

/* ---------------------------------------------------------------- fragment solver (hot path) ----- */
/* Replaces, per fragment, the body of be_func's loop -- molbe/solver.py:301-547 -- and its worker twin
 * run_solver(h1, dm0, ..., nao, nocc, n_frag, weight_and_relAO_per_center, TA, h1_e, solver, eri_file,
 * veff, veff0, ...) -> (e_f, mo_coeff, rdm1, rdm2s, rdm1_tmp), molbe/be_parallel.py:40-60, :301-307:
 * fragment RHF (helper.py:73-151) -> solve_ccsd (solver.py:829-946) -> rdm1 back-rotation (solver.py:496-505)
 * -> get_frag_energy (helper.py:220-339).  The 2-RDM is never materialised: its contraction with the
 * fragment ERIs is evaluated from t1/t2 directly (identical result).                      */
typedef struct {
  double cc_conv_tol;        /* |dE_corr|          default 1e-10 (PySCF 1e-7)                      */
  double cc_conv_tol_normt;  /* |dt|               default 1e-8  (PySCF 1e-5)                      */
  int cc_max_cycle;          /*                    default 100   (PySCF 50)                        */
  int cc_diis_space;         /*                    default 6     (PySCF 6)                         */
  double scf_conv_tol;       /* |dE_scf|           default 1e-11 (PySCF 1e-9)                      */
  double scf_conv_tol_grad;  /* ||FD-DF||          default 1e-7                                    */
  int scf_max_cycle;         /*                    default 50    (molbe/helper.py:118)             */
  int scf_diis_space;        /*                    default 8                                       */
  int warm_start;            /* reuse t1/t2 of the previous sweep as the CCSD guess (default 0)      */
  int verbose;
  int relax_density;         /* solve_ccsd(relax=True), solver.py:925-939: CCSD Lambda equations, response 1-RDM in
                              * rdm1_mo / rdm1_emb and the relaxed with_dm1=False 2-RDM in e_frag (default 0)          */
  double lambda_conv_tol;    /* |dz|               default 1e-8  (PySCF solve_lambda 1e-5)          */
  int lambda_max_cycle;      /*                    default 100                                      */
} qemb_solver_opts;
void qemb_default_opts(qemb_solver_opts* opts);

typedef void* qemb_frag_t;   /* opaque: one fragment with its ERIs resident in HBM                   */
int qemb_frag_create(int n, int n_f, qemb_frag_t* out);
int qemb_frag_free(qemb_frag_t f);
/* fragment ERIs, 4-fold packed (npair(n) x npair(n)): the dataset "f{I}" of eri_file.h5 (mbe.py:1039) */
int qemb_frag_set_eri_s4(qemb_frag_t f, const double* eri_s4_host);
int qemb_frag_set_eri_s4_dev(qemb_frag_t f, const double* eri_s4_dev);
int qemb_frag_get_eri_s4(qemb_frag_t f, double* eri_s4_host);
/* h1 = TA^T hcore TA, veff0 = TA^T V_hf TA, veff (may be NULL), centre weight and indices
 * (Frags.weight_and_relAO_per_center, pfrag.py:100)                                                  */
int qemb_frag_set_energy_data(qemb_frag_t f, const double* h1, const double* veff0, const double* veff,
                              double weight, const int* centers, int ncenter);
/* J[p,q] = (pq|rs) P[r,s], K[p,r] = (pq|rs) P[q,s] from the resident ERIs (helper.py:64 dot_eri_dm)   */
int qemb_frag_jk(qemb_frag_t f, const double* P, double* J, double* K);
/* one fragment of the sweep.  h = fock + heff (n x n); dm0 n x n or NULL; eeval: also fragment energies.
 * outputs (any may be NULL): mo_coeff n*n, mo_energy n, rdm1_emb n*n (= C rdm1 C^T / 2, Frags._rdm1),
 * rdm1_mo n*n (Frags.rdm1__), t1 o*v, t2 o*o*v*v, e_frag[3] = [e1,e2,ec], scalars.                     */
int qemb_frag_solve(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts,
                    int eeval, double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1,
                    double* t2, double* e_frag, double* e_corr_mo, double* e_scf, double* ebe_hf, int* n_iter,
                    int* scf_cycles);
/* number of Lambda iterations of the last qemb_frag_solve with relax_density (0 otherwise) */
int qemb_frag_lambda_iters(qemb_frag_t f, int* n_iter);
/* fragment RHF only: get_scfObj(fock + heff, eri, nocc, dm0) of molbe/helper.py:73-151 as used by
 * Frags.scf(fs=True) at initialisation (mbe.py:1160).  J, K: of the converged density (nullable).        */
int qemb_frag_scf(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts,
                  double* mo_coeff, double* mo_energy, double* J, double* K, double* e_scf, int* converged, int* cycles);
/* CPHF density response to npot one-body perturbations (npot x n x n) -> dPs (npot x n x n): the work inside
 * hfres_func / cphf_kernel_batch (shared/external/optqn.py:456-466, cphf_utils.py:55-81) that builds the
 * initial Jacobian of the quasi-Newton density matching.                                                 */
int qemb_frag_cphf(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts,
                   const double* vpots, int npot, double* dPs);
/* stateless one-call form (host buffers in, host buffers out) */
int qemb_ccsd_solve(int n, int nsocc, int n_f, const double* h, const double* eri_s4, const double* dm0,
                    const qemb_solver_opts* opts, const double* h1, const double* veff0, double weight,
                    const int* centers, int ncenter, double* mo_coeff, double* mo_energy, double* t1, double* t2,
                    double* rdm1_emb, double* e_frag, double* e_corr_mo, int* n_iter);
/* measurement hooks: set up SCF + integrals once, then run/timed single CCSD iterations                */
int qemb_frag_prepare_ccsd(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts);
int qemb_frag_ccsd_iterate(qemb_frag_t f, int niter, double* e_corr, double* normt);
int qemb_frag_ccsd_reset(qemb_frag_t f);
/* copy one MO-integral block of the prepared CCSD problem to the host: "oooo" "ovoo" "ovov" "ovvv" "Vl" (= (ac|bd) at
 * [a,b,c,d]) "W1base" (= ovvo[k,c,a,i] at [i,a,k,c]) "W2base" (= oovv[k,i,a,c] at [i,a,k,c]) "eo" "ev"                     */
int qemb_frag_ccsd_export(qemb_frag_t f, const char* name, double* host, int64_t nelem);

/* ---------------------------------------------------------------- AO -> fragment ERI transforms -- */
/* Dense: replaces `ao2mo.incore.full(eri_, TA, compact=True)` of BE._eri_transform "in-core"
 * (molbe/mbe.py:1035-1039).  The AO tensor is uploaded once per system and stays resident.            */
typedef void* qemb_aoeri_t;
int qemb_aoeri_upload(int N, const double* eri, int sym /* 8, 4 or 1 */, qemb_aoeri_t* out);
int qemb_aoeri_free(qemb_aoeri_t ao);
/* TA: N x n (host).  Result 4-fold packed (npair(n) x npair(n)) to out_s4_host (nullable) and/or
 * straight into a fragment handle (nullable) -- the HDF5 dataset "f{I}" hand-off without the disk.     */
int qemb_ao2mo_dense(qemb_aoeri_t ao, const double* TA, int n, double* out_s4_host, qemb_frag_t frag);

/* Density fitted: replaces integral_direct_DF (molbe/eri_onthefly.py:45-145, "int-direct-DF") and the
 * transform_integral / transform_integral_cuda pair injected into _run_sparse_df_driver
 * (molbe/eri_sparse_DF.py:535-556, :677-678, :701-702; C++ _cpp/eri_sparse_DF.cpp:724-751).
 * qemb_df_create factors (P|Q) on the device (eri_onthefly.py:108); qemb_lpq_upload takes an existing
 * lower Cholesky factor instead (the `build_lowtri_PQ` seam / GPU_MatrixHandle, eri_sparse_DF.cpp:64-107). */
typedef void* qemb_df_t;
int qemb_df_create(int naux, const double* j2c, qemb_df_t* out);
int qemb_lpq_upload(const double* L_PQ, int naux, qemb_df_t* out);
int qemb_df_free(qemb_df_t df);
/* 3-index integrals: layout 0 = (N,N,naux) "pqL" as getints3c returns them (eri_onthefly.py:85),
 * 1 = (naux,N,N), 2 = (naux, npair(N)) unique pairs mu >= nu (SemiSparseSym3DTensor without screening) */
int qemb_df_set_ints(qemb_df_t df, int N, const double* ints, int layout);
/* The reference's SemiSparseSym3DTensor itself (_cpp/eri_sparse_DF.cpp:110-298), never expanded to the dense (P|mu nu):
 * unique_dense_data = the (naux x n_unique) column-major matrix of the reference (one aux vector per stored unique AO pair;
 * n_unique x naux when read row-major); exch_reachable_with_offsets (:260-279) in CSR form: the partners of AO mu are
 * reach_nu[reach_ptr[mu] .. reach_ptr[mu+1]) and reach_off[...] is the column of the pair's aux vector.  Device memory is
 * O(n_unique naux); both transforms below then run the screened algorithm on this storage.                            */
int qemb_df_set_ints_semisparse(qemb_df_t df, int N, int64_t n_unique, const double* unique_dense_data, const int64_t* reach_ptr,
                                const int32_t* reach_nu, const int64_t* reach_off);
int qemb_df_transform(qemb_df_t df, const double* TA, int n, double* out_s4_host, qemb_frag_t frag);
/* the same with the MO-coefficient screening of transform_integral(int_P_mu_nu, TA, S_abs, L_PQ, MO_coeff_epsilon)
 * (_cpp/eri_sparse_DF.cpp:739-751): (P|mu i) is kept only for mu with |S_abs TA|(mu,i) >= epsilon (get_AO_per_MO :443);
 * unstored (screened) AO pairs of the SemiSparseSym3DTensor are passed as zeros of the packed layout 2.              */
int qemb_df_transform_screened(qemb_df_t df, const double* TA, int n, const double* S_abs, double MO_coeff_epsilon,
                               double* out_s4_host, qemb_frag_t frag);

/* ---------------------------------------------------------------- Schmidt decomposition ---------- */
/* schmidt_decomposition(mo_coeff, nocc, AO_in_frag, thr_bath) -> (TA_lo_eo, n_f, n_b), molbe/pfrag.py:403-411.
 * lmo: N x nmo row-major; TA_lo_eo: caller buffer N x ld (ld >= n_f + n_b; 2*n_f always suffices).      */
int qemb_schmidt(const double* lmo, int N, int nmo, int nocc, const int64_t* frag_idx, int n_f, double thr,
                 double* TA_lo_eo, int ld, int* n_b, int* sweeps);
/* Same contract and (for the idempotent HF 1-RDM of pfrag.py:448-450) the same bath as qemb_schmidt, through the
 * rank-n_f invariant subspace spanned by D[env,frag]: O(N_env n_f nocc) instead of the O(N_env^3) eigenproblem.   */
int qemb_schmidt_subspace(const double* lmo, int N, int nmo, int nocc, const int64_t* frag_idx, int n_f, double thr,
                          double* TA_lo_eo, int ld, int* n_b, int* sweeps);
/* schmidt_decomp_svd(rdm, Frag_sites, thr_bath) -> TA, kbe/solver.py:9 (real part)                     */
int qemb_schmidt_svd(const double* rdm, int N, const int64_t* frag_idx, int n_f, double thr, double* TA, int ld,
                     int* n_b, int* sweeps);
/* Frags.get_nsocc (molbe/pfrag.py:208-239): Cproj = TA^T S C_occ (n x nocc) -> P (n x n, nullable),
 * nsocc, initial fragment MOs (n x n)                                                                  */
int qemb_nsocc_guess(const double* Cproj, int n, int nocc, double* P, int* nsocc, double* mo_coeffs);
/* plain host-in/host-out matrix product on the device: C(MxN) = op(A) op(B) (convenience for TA = W @ TA_lo_eo) */
int qemb_matmul(int64_t M, int64_t N, int64_t K, const double* A, int transA, const double* B, int transB, double* C);

#ifdef __cplusplus
}
#endif
#endif /* QEMB_HIP_H */
