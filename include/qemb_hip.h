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
 *   - This header is the PRODUCT ABI (what a QuEmb binding calls).  The device primitives the drivers are composed of,
 *     the device timers and the measurement / tuning hooks used by tests/, bench.py and tools/ are declared in qemb_hip_ops.h.
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
#define QEMB_WARN_NOCONV 1       /* only with strict_convergence = 0: results returned, but a solve did not converge */

/* ---------------------------------------------------------------- library / device ------------- */
int qemb_init(int device);                 /* select the GPU, create the library stream            */
const char* qemb_last_error(void);
const char* qemb_backend(void);            /* "hip-gfx950"                                         */
int qemb_sync(void);                      /* the calling thread's stream (execution context)      */
int qemb_device_sync(void);               /* every stream of the device                           */
int qemb_mem_info(size_t* free_bytes, size_t* total_bytes);

/* raw device buffers (for callers that keep tensors resident, e.g. bench.py / multi-fragment sweeps) */
int qemb_malloc(void** dptr, size_t bytes);
int qemb_free(void* dptr);                 /* parks the block for reuse (caching allocator)       */
int qemb_trim(void);                       /* hand every parked block back to the driver          */
int qemb_trim_all(void);              /* the same for EVERY execution context (their streams are drained first): between phases with very different working sets */
int qemb_h2d(void* dptr, const void* host, size_t bytes);
int qemb_d2h(void* host, const void* dptr, size_t bytes);
/* upload that only ORDERS the copy on the calling context's stream (for data that context consumes; qemb_sync or any later download of the context completes it).  The host
 * buffer is free on return: copies of up to 256 KB leave from pinned slots of the context (round 5 -- the runtime's pageable path serialised the host threads of a
 * batched sweep), larger ones fall back to the waiting path of qemb_h2d. */
int qemb_h2d_async(void* dptr, const void* host, size_t bytes);
int qemb_d2d(void* dst, const void* src, size_t bytes);

/* Execution contexts (one HIP stream + workspaces + block cache each; no reference counterpart -- the reference overlaps
 * fragments with a process pool, be_parallel.py:484).  qemb_ctx_count(n) makes contexts 0..n-1 available (0 = default) and
 * returns how many exist (or < 0); qemb_ctx_bind(k) binds the CALLING host thread to context k, so that several host
 * threads can each drive a fragment on their own stream (device timers of a context: qemb_ctx_timer_read, qemb_hip_ops.h).
 * qemb_ctx_partition(parts): contexts 1, 2, ... are spread over `parts` disjoint, interleaved sets of compute units (0 / 1: every context on the
 * whole chip), so that the HBM-bound passes of one large fragment run beside the MFMA-bound products of another; existing contexts are drained and get
 * new streams -- call it between sweeps, not while other threads are solving. */
int qemb_ctx_count(int n);
int qemb_ctx_bind(int k);
int qemb_ctx_partition(int parts);

/* ---------------------------------------------------------------- fragment solver (hot path) ----- */
/* Replaces, per fragment, the body of be_func's loop -- molbe/solver.py:301-547 -- and its worker twin
 * run_solver(h1, dm0, ..., nao, nocc, n_frag, weight_and_relAO_per_center, TA, h1_e, solver, eri_file,
 * veff, veff0, ...) -> (e_f, mo_coeff, rdm1, rdm2s, rdm1_tmp), molbe/be_parallel.py:40-60, :301-307:
 * fragment RHF (helper.py:73-151) -> solve_ccsd (solver.py:829-946) -> rdm1 back-rotation (solver.py:496-505)
 * -> get_frag_energy (helper.py:220-339).  The 2-RDM is never materialised: its contraction with the
 * fragment ERIs is evaluated from t1/t2 directly (identical result).                      */
typedef struct {
  uint32_t struct_size;      /* sizeof(qemb_solver_opts) of the header the caller was built against; qemb_default_opts sets it and
                              * every entry point that takes options rejects another value (QEMB_ERR_ARG): a binding whose field
                              * list has drifted from this header fails loudly instead of reading flags out of padding           */
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
  int strict_convergence;    /* 1 (default): a fragment RHF / CCSD / Lambda solve that does not converge is an error, status
                              * QEMB_ERR_NOCONV, no outputs.  0: the reference's behaviour -- PySCF warns and carries on with what it
                              * has (helper.py:128-149, solver.py:905-912): every output is filled from the unconverged state and the
                              * call returns QEMB_WARN_NOCONV (> 0); qemb_last_error() says which solve it was.            */
} qemb_solver_opts;
void qemb_default_opts(qemb_solver_opts* opts);   /* always start from this; then change single fields */

typedef void* qemb_frag_t;   /* opaque: one fragment with its ERIs resident in HBM                   */
int qemb_frag_create(int n, int n_f, qemb_frag_t* out);
int qemb_frag_free(qemb_frag_t f);
/* fragment ERIs, 4-fold packed (npair(n) x npair(n)): the dataset "f{I}" of eri_file.h5 (mbe.py:1039) */
int qemb_frag_set_eri_s4(qemb_frag_t f, const double* eri_s4_host);
int qemb_frag_set_eri_s4_dev(qemb_frag_t f, const double* eri_s4_dev);
int qemb_frag_get_eri_s4(qemb_frag_t f, double* eri_s4_host);
/* Optional: the fragment's fitted 3-index factor B[naux][npair(n)] (rows L, unique pairs i >= j), the `bb` of
 * integral_direct_DF whose product `bb.T @ bb` IS the block above (molbe/eri_onthefly.py:141-143).  With it a solve forms
 * its MO-basis integrals from the factor -- transformed with the fragment's orbitals and multiplied with itself,
 * 2 naux npair(n)^2 flops -- instead of the four quarter transformations of the packed block (PySCF's ao2mo inside
 * cc.CCSD(...).ao2mo(), molbe/solver.py:900), while that is the cheaper route (naux <= 8 n); the results agree to
 * rounding.  Set it AFTER the ERIs it belongs to: new ERIs drop it, and a factor whose product differs from the resident block (random probe of the
 * WHOLE block: B^T (B x) against eri_s4 x for two vectors x, 1e-9 relative) is refused with QEMB_ERR_ARG and no factor is kept.  qemb_df_transform(..., frag) hands it over itself.
 * qemb_frag_mo_route: -1 choose by cost (default), 0 always the four-index transformation, 1 always the factor.        */
int qemb_frag_set_df_factor(qemb_frag_t f, int naux, const double* B_host);
int qemb_frag_set_df_factor_dev(qemb_frag_t f, int naux, const double* B_dev);
int qemb_frag_mo_route(qemb_frag_t f, int route);
/* A fragment that LIVES on its 3-index factor (round 5): qemb_frag_set_df_only[_dev] sets B (naux x npair(n), eri = B^T B) and drops any resident
 * 4-fold packed block -- 8 naux npair bytes resident instead of 8 npair^2 (128 MB instead of 4.7 GB at n = 220, naux = 660).  J / K of the fragment RHF
 * (helper.py:28-69 get_veff: J = B^T (B Dp), K = sum_L (B_L Co)(B_L Co)^T), the MO integrals, energies, relaxed densities and the CPHF response come
 * from the factor; qemb_frag_jk and qemb_frag_get_eri_s4 keep working (the block is formed for that call: B^T B).  qemb_frag_mo_route(f, 0) still forces
 * the four-index transformation (the block is then a transient of each solve).  Any qemb_frag_set_eri_s4 ends the mode.
 * qemb_frag_resident_bytes: device bytes the fragment keeps between solves (ERIs and / or factor, orbitals, densities, kept amplitudes).            */
int qemb_frag_set_df_only(qemb_frag_t f, int naux, const double* B_host);
int qemb_frag_set_df_only_dev(qemb_frag_t f, int naux, const double* B_dev);
int qemb_frag_resident_bytes(qemb_frag_t f, int64_t* bytes);
int qemb_frag_mo_route_used(qemb_frag_t f, int* used_factor, int* naux);   /* what the last solve did; naux of the factor held (0: none) */
/* h1 = TA^T hcore TA, veff0 = TA^T V_hf TA, veff (may be NULL), centre weight and indices
 * (Frags.weight_and_relAO_per_center, pfrag.py:100)                                                  */
int qemb_frag_set_energy_data(qemb_frag_t f, const double* h1, const double* veff0, const double* veff,
                              double weight, const int* centers, int ncenter);
/* J[p,q] = (pq|rs) P[r,s], K[p,r] = (pq|rs) P[q,s] from the resident ERIs (helper.py:64 dot_eri_dm)   */
int qemb_frag_jk(qemb_frag_t f, const double* P, double* J, double* K);
/* one fragment of the sweep.  h = fock + heff (n x n); dm0 n x n or NULL; eeval: also fragment energies.
 * outputs (any may be NULL): mo_coeff n*n, mo_energy n, rdm1_emb n*n (= C rdm1 C^T / 2, Frags._rdm1),
 * rdm1_mo n*n (Frags.rdm1__), t1 o*v, t2 o*o*v*v, e_frag[3] = [e1,e2,ec], scalars.  0 < nsocc <= n; nsocc == n (no virtual
 * orbitals) returns the mean-field results with E_corr = 0 and empty amplitudes, as PySCF's CCSD does.   */
int qemb_frag_solve(qemb_frag_t f, int nsocc, const double* h, const double* dm0, const qemb_solver_opts* opts,
                    int eeval, double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1,
                    double* t2, double* e_frag, double* e_corr_mo, double* e_scf, double* ebe_hf, int* n_iter,
                    int* scf_cycles);
/* Every fragment of a sweep in ONE call -- the "batched variant over a list of fragments per device" of the solver seam (the
 * reference's pool of workers, molbe/be_parallel.py:484-517).  Small fragments (octane BE2: six fragments of ~40 orbitals) are bound by the
 * NUMBER of dependent kernel launches, ~110 per CCSD iteration at 4-5 us each whatever the size: here the fragment RHF, the MO transformation
 * and the density / energy evaluation run per fragment on one stream each, and the CCSD iterations of all fragments run in lock step -- every
 * operation of the amplitude update is ONE grouped launch over all fragments still iterating.  Each fragment performs exactly the operations of
 * qemb_frag_solve in the same order: results are bit-identical.  Arguments are arrays over the fragments (pointer arrays and their entries may
 * be NULL where qemb_frag_solve allows NULL); e_frag is 3 * nfrag.  stats (nullable, 5 values): merged launch sequences run, launches they
 * issued, of which grouped, recorded operations they covered, largest number of fragments in one sequence.                               */
int qemb_frag_solve_batch(int nfrag, const qemb_frag_t* frags, const int* nsocc, const double* const* h, const double* const* dm0,
                          const qemb_solver_opts* opts, int eeval, double* const* mo_coeff, double* const* mo_energy,
                          double* const* rdm1_emb, double* const* rdm1_mo, double* const* t1, double* const* t2, double* e_frag,
                          double* e_corr_mo, double* e_scf, double* ebe_hf, int* n_iter, int* scf_cycles, int64_t* stats);
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
/* ---------------------------------------------------------------- multi-GPU exchange -------------- */
/* One process per GPU; fragments are sharded over the ranks and the ONLY exchange of a sweep is one all-reduce of the residual buffer
 * [edge values, centre values, sum centre diag, e1, e2, ec, n_iter, failure flag] -- what be_func_parallel gets back from its pathos pool
 * as pickled result tuples (molbe/be_parallel.py:484-517) before solve_error reads Fobjs[j]._rdm1 (molbe/solver.py:683-778); SURVEY.md 8(e).
 * The communicator is RCCL (xGMI inside a node), created once per process on the device and stream of qemb_init, and needs no Python
 * package: qemb_comm_unique_id on ONE rank -> the 128 bytes travel to every rank out of band (file, socket, MPI_Bcast, a torch store...)
 * -> qemb_comm_init(rank, world, id) on every rank (collective).  Buffers are HOST buffers, reduced in place; every rank receives the
 * bit-identical result.  Without qemb_comm_init the process is a world of one: info reports (0, 1), all-reduce fails.                  */
#define QEMB_COMM_ID_BYTES 128
#define QEMB_COMM_SUM 0
#define QEMB_COMM_MAX 1
int qemb_comm_unique_id(void* id_out /* QEMB_COMM_ID_BYTES */);
int qemb_comm_init(int rank, int world, const void* id);
int qemb_comm_info(int* rank, int* world);
int qemb_comm_allreduce(double* host_buf, int64_t n, int op);
int qemb_comm_destroy(void);

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
/* The same two transforms handing the FITTED FACTOR ALONE to the fragment (bb of eri_onthefly.py:141; the bb^T bb product of :143 is not formed): the
 * fragment then lives on it (qemb_frag_set_df_only semantics) -- the a4 transform without its 2 naux npair^2 flops and without the 8 npair^2 bytes.   */
int qemb_df_transform_factor(qemb_df_t df, const double* TA, int n, qemb_frag_t frag);
int qemb_df_transform_screened_factor(qemb_df_t df, const double* TA, int n, const double* S_abs, double MO_coeff_epsilon, qemb_frag_t frag);

/* Gamma-point periodic (CC-GDF) variant of the direct DF transform, kbe/eri_onthefly.py:48-241.
 * qemb_df_create_pbc: _j2c_cholesky_or_eig (:19-45) -- the periodic metric may be indefinite: Cholesky when it succeeds (*ischol = 1),
 *   otherwise the fit matrix V d^{-1/2} V^T over the eigenvalues d > 1e-14 (*ischol = 0); bb = L^{-1} b or bb = fit b (:222-227).
 * qemb_df_alloc_ints: zeroed fitted tensor (L|mu nu), real and imaginary part (`pqL_frag`, :152-155, held once at the AO level:
 *   TA^T [sum_G F (G|mu nu)] TA = sum_G F (TA^T (G|mu nu) TA), so the per-fragment transform of every plane-wave block is not needed).
 * qemb_df_add_pw_block: += sum_G F[L,G] (G|mu nu) for nG plane waves (:176-199); F = ft_ao(chgcell, Gv)^H as naux x nG (re, im),
 *   (G|mu nu) = ft_aopair * coulG^* as nG x N x N (re, im).
 * qemb_df_add_rs_block: rows [p0, p1) += the real-space block (aux_e2(auxcell) - aux_e2(chgcell), :85-103, :201-217) as (p1-p0) x N x N.
 * qemb_df_pw_imag_absmax: max |Im (L|mu nu)| -- zero for a +-G symmetric mesh; the host mirror reproduces the reference's
 *   `Imaginary part of ERI is larger than 1e-6` check (:231-236) from it.   qemb_df_pw_select: the tensor qemb_df_transform reads
 *   (0 = real part, 1 = imaginary part, 2 = their sum: Re / Im of bb^T bb follow from three real transforms).                      */
int qemb_df_create_pbc(int naux, const double* j2c, qemb_df_t* out, int* ischol);
int qemb_df_alloc_ints(qemb_df_t df, int N);
int qemb_df_add_pw_block(qemb_df_t df, int nG, const double* F_re, const double* F_im, const double* pw_re, const double* pw_im);
int qemb_df_add_rs_block(qemb_df_t df, int p0, int p1, const double* block);
int qemb_df_pw_imag_absmax(qemb_df_t df, double* out);
int qemb_df_pw_select(qemb_df_t df, int part);

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
/* ---------------------------------------------------------------- AO screening (semi-sparse DF) --- */
/* int |chi_a| |chi_b| of unnormalised uncontracted Cartesian Gaussians by Gauss-Hermite quadrature: the primitive stage of
 * approx_S_abs (molbe/eri_sparse_DF.py:733-865, :928-959; numba on the host in the reference).  nsh primitive shells (l <= 4, exponent,
 * centre xyz[3s..], first Cartesian function cart0[s], components in libcint order); roots / weights of the nroots-point rule;
 * out: ncart x ncart (host).  The contraction to |c|^T s |c| and the reachability lists are host logic (eri_sparse_DF.py).           */
int qemb_abs_overlap_prim(int nsh, const int* l, const double* ex, const double* xyz, const int64_t* cart0, int64_t ncart, int nroots,
                          const double* roots, const double* weights, double* out);
/* plain host-in/host-out matrix product on the device: C(MxN) = op(A) op(B) (convenience for TA = W @ TA_lo_eo) */
int qemb_matmul(int64_t M, int64_t N, int64_t K, const double* A, int transA, const double* B, int transB, double* C);

#ifdef __cplusplus
}
#endif
#endif /* QEMB_HIP_H */
