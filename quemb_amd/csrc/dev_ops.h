// dev_ops.h -- the device-operation layer of libqemb_hip.
//
// Every driver in this directory (ccsd.cpp, scf.cpp, ao2mo.cpp, schmidt.cpp, api.cpp) is plain
// C++ written against THIS interface only.  The product library links dev_ops_hip.hip (hand-written
// gfx950 kernels).  tests/hostcheck/ links the same drivers against a slow scalar mock so that the
// host logic can be exercised in a GPU-less container; that mock is test infrastructure and is never
// part of libqemb_hip.so.
//
// Conventions: FP64 only, row-major, all pointers are DEVICE pointers unless the name says host,
// sizes are int64_t.  Every function returns 0 on success, <0 on error (qemb::last_error() has text).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include "../../include/qemb_hip_ops.h"

namespace qemb {

// ---- error channel -------------------------------------------------------------------------
void set_error(const std::string& msg);
const char* last_error();

// status codes: QEMB_OK / QEMB_ERR_* from the public header


// ---- device / memory -----------------------------------------------------------------------
int dev_init(int device);            // select device, create the library stream
int dev_sync();                      // wait for the calling thread's stream
int dev_sync_device();               // wait for every stream of the device (hipDeviceSynchronize)
// Execution contexts: one HIP stream + workspaces + block cache each.  dev_ctx_count(n) makes contexts 0..n-1 available
// (0 = default) and returns how many exist; dev_ctx_bind(k) binds the CALLING host thread to context k.
int dev_ctx_count(int n);
int dev_ctx_bind(int k);
int dev_ctx_partition(int parts);
int dev_alloc(void** p, size_t bytes);
int dev_free(void* p);                // parks the block in a free list (see dev_trim)
int dev_trim();                       // release every parked block back to the driver (the calling context's)
int dev_trim_all();                   // ... of EVERY context (each context's stream is drained first): between phases of very different working sets
int dev_h2d(void* dst, const void* src_host, size_t bytes);
int dev_h2d_async(void* dst, const void* src, size_t bytes);      // ordered on the calling context's stream only (the source is free on return)
int dev_d2h(void* dst_host, const void* src, size_t bytes);
// device -> PINNED host memory without waiting: the data are there after the next dev_sync of the calling context (lock-step sweeps read
// the scalars of several fragments' streams with one wait each instead of one per transfer)
int dev_d2h_async(void* dst_pinned, const void* src, size_t bytes);
int dev_pinned_alloc(void** p, size_t bytes);
int dev_pinned_free(void* p);
int dev_d2d(void* dst, const void* src, size_t bytes);
int dev_fill(double* x, int64_t n, double value);
int dev_mem_info(size_t* free_b, size_t* total_b);
// measurement hook: real driver allocations (pool misses) since the last reset -- their number, bytes and the host time spent in them
int dev_alloc_stats(long long* n_driver_allocs, long long* n_driver_frees, double* ms_in_driver_calls, double* gb_allocated, int reset);
const char* dev_backend_name();      // "hip-gfx950" for the product, "hostcheck" for the mock

// ---- stream capture (hipGraph) of launch-bound loops ------------------------------------------------------------------
// Between begin and end every dev_* launch is recorded instead of executed; the recorded graph can then be replayed
// with one host call.  Only pure launch sequences may be captured (no allocation, no host transfer, no timers).
// dev_graph_begin returns 1 (not an error) when the backend cannot capture; the caller then simply runs eagerly.
typedef void* dev_graph_t;
int dev_graph_begin(int for_tape = 0);       // for_tape: the capture will end as a tape (dev_tape_end): parallel regions leave their markers
// Parallel regions inside a tape capture: chains of operations that do not depend on each other (dev_ops_hip.hip has the contract).  No-ops when the calls are
// executed or captured for an executable graph: program order is always a valid order.
int dev_region_begin();
int dev_region_chain();
int dev_region_end();
int dev_graph_end(dev_graph_t* out);
int dev_graph_launch(dev_graph_t g);
int dev_graph_destroy(dev_graph_t g);
bool dev_capturing();
// A capture can also end as a TAPE: the recorded launch sequence kept as data (kernel, grid, arguments; copies) instead of an executable
// graph, so that the tapes of SEVERAL fragments can be executed together on one stream, position by position -- launches of the same kernel
// at the same position become ONE grouped launch over all fragments (dev_tape_run).  Small fragments are bound by the number of dependent
// launches (4-5 us each whatever their size): F fragments in lock step cost the launches of one.  dev_tape_end returns 1 (not an error) when the
// backend cannot tape; the caller then keeps the per-fragment path.
typedef void* dev_tape_t;
int dev_tape_end(dev_tape_t* out);
int dev_tape_run(const dev_tape_t* tapes, int n);          // on the calling thread's stream; the tapes stay valid and may be run again
int dev_tape_destroy(dev_tape_t t);
// 1 when two tapes are the same launch sequence (kernels, grids and every argument byte of the kernels the grouped launches know; 0 otherwise) -- the check of
// the tape cache (fragment.cpp, QEMB_TAPE_CACHE_CHECK)
int dev_tape_equal(dev_tape_t a, dev_tape_t b);
// A tape is valid as long as every buffer its launches name is where it was.  The caller brackets the allocations of a solve with these two: the value returned
// is a hash of every block dev_alloc handed out on the calling thread's context in between (size and address, in order) and of the context's scratch blocks --
// equal values on two solves mean the same buffers at the same places (round 5: a fragment's tape is kept from sweep to sweep under that key)
void dev_alloc_trace_begin();
unsigned long long dev_alloc_trace_end();
// counters of the calling thread's last dev_tape_run: launches issued, of which grouped, recorded operations covered
int dev_tape_last_stats(long long* launches, long long* grouped, long long* operations);

// ---- timing (HIP events on the library stream) ------------------------------------------------
// A "lap" accumulates elapsed device time of every region bracketed with the same slot id.
int dev_timer_begin(int slot);
int dev_timer_end(int slot);
int dev_timer_read(int slot, double* total_ms, int64_t* count);   // syncs
int dev_timer_reset(int slot);
int dev_ctx_timer_read(int ctx, int slot, double* total_ms, int64_t* count, int reset);   // another context's timers (idle context)
int dev_timer_live_events(int slot);   // event pairs currently held by the calling context's slot (bounded; test hook)
enum { TIMER_LADDER = 0, TIMER_RINGS = 1, TIMER_ITER = 2, TIMER_AO2MO = 3, TIMER_SCF = 4,
       TIMER_GEMM_ANY = 5, TIMER_SCHMIDT = 6, TIMER_DF = 7, TIMER_NSLOTS = 16 };
// A lap that always ends: the stop event is recorded when the scope is left, also on an early (error) return of the bracketed region.
struct TimerScope {
  int slot; bool open;
  explicit TimerScope(int s) : slot(s), open(dev_timer_begin(s) == 0) {}
  int close() { if (!open) return 0; open = false; return dev_timer_end(slot); }
  ~TimerScope() { if (open) (void)dev_timer_end(slot); }
  TimerScope(const TimerScope&) = delete;
  TimerScope& operator=(const TimerScope&) = delete;
};

// ---- GEMM (FP64 MFMA) --------------------------------------------------------------------------
// C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b],   b = 0..batch-1
//   op(A) is M x K.  a_kcontig: A(m,k) = A[m*lda + k]   (row-major M x K);
//                   !a_kcontig: A(m,k) = A[k*lda + m]   (stored K x M, i.e. "transposed")
//   op(B) is K x N.  b_kcontig: B(k,n) = B[n*ldb + k]   (stored N x K, i.e. "transposed");
//                   !b_kcontig: B(k,n) = B[k*ldb + n]   (row-major K x N)
//   C is row-major M x N with leading dimension ldc.  Batch strides are in elements.
struct GemmDesc {
  int64_t M, N, K;
  double alpha, beta;
  const double* A; int64_t lda; int a_kcontig; int64_t strideA;
  const double* B; int64_t ldb; int b_kcontig; int64_t strideB;
  double* C; int64_t ldc; int64_t strideC;
  int64_t batch;
  int cfg = -1;        // tile configuration override (-1: automatic)
  int ksplit = 0;      // split-K override (0: automatic, < 0: never)
  // keep_slabs (needs ksplit > 1, alpha = 1, beta = 0, batch = 1): the K slices' partial products stay where they are -- C receives
  // gemm_slab_count(K, ksplit) slabs [M][N] (ld = N, ldc ignored) and the consumer adds them up in slab order (what the reduction pass would do)
  int keep_slabs = 0;
  // Slab-aware rows of a !a_kcontig A operand (round 5): row m sits a_slab_skip * (m / a_slab) elements further on, A(m,k) = A[k*lda + m + (m / a_slab) * a_slab_skip].
  // With a_slab = lda = n and a_slab_skip = n^2 - n the rows (pair, q) of a stack of n x n slabs X[pair][k][q] are ONE tall operand: the batched product
  // C^T . slab of the MO transformation runs flat on the tall tile.  0: plain rows.  (a_slab even when the operand is loaded 16 bytes at a time.)
  int64_t a_slab = 0, a_slab_skip = 0;
};
inline int gemm_slab_count(int64_t K, int ksplit) {      // the K slices dev_gemm cuts for an explicit ksplit (32-aligned chunks)
  if (ksplit <= 1) return 1;
  int64_t chunk = (K + ksplit - 1) / ksplit;
  chunk = (chunk + 31) / 32 * 32;
  const int64_t S = (K + chunk - 1) / chunk;
  return S > 1 ? (int)S : 1;
}
int dev_gemm(const GemmDesc& g);
// measurement hook: 2 M N K batch of every product issued (or recorded into a capture) by ANY host thread since the last reset -- the executed flops the size
// sweep of bench.py divides by the iteration time (a captured update replayed k times counts once: read it around eager iterations)
int dev_gemm_flop_count(double* flops, int reset);
// test / tuning hooks of the calling host thread: force a tile configuration (-1: automatic), switch the automatic split-K off
void dev_gemm_set_force_cfg(int cfg);
void dev_gemm_set_auto_splitk(int enabled);
// hint of the calling host thread: this many products of each shape will be launched together (grouped launches of a lock-step sweep); read by callers
// that choose a tile themselves (the ring products of small fragments)
void dev_gemm_set_peers(int n);
int dev_gemm_peers();
// one product timed on its own with the sustained shader clock read back (see gemm_f64.hip); syncs the stream -- a measuring aid
int dev_gemm_probe(const GemmDesc& g, double* ms, double* ghz, long long* workgroups);
// diagnostic tile configurations (3xx): per-wave s_memtime sums around the per-tile barrier, averaged over the waves of one launch
int dev_gemm_stamps(const GemmDesc& g, int waves_per_wg, double* out7);

// ---- strided tensor copy / add (up to 4 dims) -------------------------------------------------
// out[i0*so[0]+i1*so[1]+i2*so[2]+i3*so[3]] = alpha * in[i0*si[0]+...+i3*si[3]] + beta * out[...]
// for 0 <= ik < dim[k].  Covers permutations, block extraction/placement, axpby, scaling.
// beta == 0 never reads out.
struct Copy4Desc {
  int64_t dim[4];
  const double* in; int64_t si[4];
  double* out; int64_t so[4];
  double alpha, beta;
  const double* base = nullptr;   // out = alpha*in + beta*base, base addressed like out (nullptr: base = out, i.e. accumulate in place)
  // optional second output of the same pass, addressed like out:  out2 = c2a * in2 + c2b * (the value written to out)
  double* out2 = nullptr; const double* in2 = nullptr; double c2a = 0.0, c2b = 0.0;
};
int dev_copy4(const Copy4Desc& c);

// out[i0,i1,i2,i3] (strides so) = beta*out + alpha * u[i0*su0 + i2*su2] * v[i1*sv1 + i3*sv3]
// (the t1 (x) t1 outer products of tau and the disconnected RDM pieces)
struct Outer4Desc {
  int64_t dim[4];
  const double* u; int64_t su0, su2;
  const double* v; int64_t sv1, sv3;
  double* out; int64_t so[4];
  double alpha, beta;
  const double* base = nullptr;   // out = alpha*u(x)v + beta*base, base addressed like out (nullptr: base = out)
};
int dev_outer4(const Outer4Desc& c);
// out = sum_{k < nterms} coef[k] * xs[k] + beta * out over n contiguous elements, one pass (nterms <= 8; out may alias any xs[k])
int dev_lincomb(int64_t n, int nterms, const double* coef, const double* const* xs, double beta, double* out);

// x[i0,i1,i2,i3] (contiguous, dims d0..d3) *= 1 / (ea[i0] + eb[i1] - ec[i2] - ed[i3])
// (orbital-energy denominators; pass d1 = d3 = 1 with eb = ed = nullptr for t1)
int dev_div_denom(double* x, int64_t d0, int64_t d1, int64_t d2, int64_t d3,
                  const double* ea, const double* eb, const double* ec, const double* ed);

// A[r][c] = A[c][r] for r < c  (n x n, leading dimension lda): completes a matrix whose lower triangle was computed
int dev_mirror_lower(int64_t n, double* A, int64_t lda);

// K[p,r] = sum_{q,s} (pq|rs) D[q,s] from the half-unpacked tensor H[P(p,q)][r][s] (p >= q rows only, npair(n) x n x n):
// row (p,q) feeds K[p,:] with D[q,:] and, for p != q, K[q,:] with D[p,:].  Deterministic (per-row partials, fixed-order sum).
int dev_k_from_pairs(int64_t n, const double* H, const double* D, double* K);
// The same K and, in the same pass, the packed Coulomb vector Jp[P(p,q)] = sum_{r>=s} (pq|rs) Dp[P(r,s)] from the 4-fold packed block
// S4[P(p,q)][P(r,s)] (half the bytes of H; Dp = D + D^T off the diagonal, D on it, packed).  Jp and Dp may be null (K only).  n <= 1024.
int dev_jk_from_packed(int64_t n, const double* S4, const double* D, const double* Dp, double* Jp, double* K);

// ---- pair-packed MO transformation helpers ---------------------------------------------------------------------------
// out[P(x,y), c] = in[(x*n + y), c] for x >= y  (row gather of an (n*n) x ncols matrix; ncols-long rows)
int dev_pack_pair_rows(int64_t n, int64_t ncols, const double* in, double* out);
// ---- gathers from the PAIR-FIRST MO tensor Mp[P(p,q)][r][s] = (pq|rs) (npair(n) x n x n), all reads in contiguous runs
// out (contiguous sp x sq x sr x ss) = Mp[P(p0+p, q0+q)][r0+r][s0+s]
int dev_extract_pf(int64_t n, const double* Mp, int64_t p0, int64_t q0, int64_t r0, int64_t s0, int64_t sp, int64_t sq,
                   int64_t sr, int64_t ss, double* out);
// T: [npair(n)][n][n] with T[P(r,s)][c][x] -> out[sx][sr][ss][sc] = T[P(r0+r, s0+s)][c0+c][x0+x]   (3/4-transformed integrals)
// (slab: doubles per pair in T, rows of n; 0 = n * n.  The factor route of mo_transform keeps only the first nf rows of every slab.)
int dev_extract_pf_t(int64_t n, const double* T, int64_t x0, int64_t r0, int64_t s0, int64_t c0, int64_t sx, int64_t sr,
                     int64_t ss, int64_t sc, double* out, int64_t slab = 0);
// (+/-) ladder operands: Vp[P(ab),P(cd)] = Mp[P(va,vc)][vb][vd] + Mp[P(vb,vc)][va][vd]  (v* = o + *), Vm with the minus sign
int dev_ladder_pack_vvvv_pf(int64_t n, int64_t o, const double* Mp, double* Vp, int64_t ldp, double* Vm, int64_t ldm);

// ---- (+/-) packed pp-ladder: R_ijab = sum_cd (ac|bd) tau_ijcd through symmetric / antisymmetric pair combinations ------
// pairs: P(x,y) = x(x+1)/2 + y for x >= y ("plus" blocks), Q(x,y) = x(x-1)/2 + y for x > y ("minus" blocks).
// Vp[P(a,b), P(c,d)] = (ac|bd) + (ad|bc),  Vm[Q(a,b), Q(c,d)] = (ac|bd) - (ad|bc)   from the MO tensor M[p,q,r,s] (n^4),
// virtual indices offset by o; row strides ldp / ldm (>= number of pairs, padding columns are zero-filled).
int dev_ladder_pack_vvvv(int64_t n, int64_t o, const double* M, double* Vp, int64_t ldp, double* Vm, int64_t ldm);
// Tp[P(i,j), P(c,d)] = w_cd (tau_ijcd + tau_ijdc), w = 1/2 (c > d) or 1/4 (c == d);  Tm[Q(i,j), Q(c,d)] = (tau_ijcd - tau_ijdc)/2
int dev_ladder_pack_tau(int64_t o, int64_t v, const double* tau, double* Tp, int64_t ldp, double* Tm, int64_t ldm);
// with Rp[P(i,j), P(a,b)], Rm[Q(i,j), Q(a,b)]:  t2[i,j,a,b] += Rp + Rm, t2[i,j,b,a] += Rp - Rm, t2[j,i,a,b] += Rp - Rm,
// t2[j,i,b,a] += Rp + Rm  (each distinct element once; Rm = 0 where i == j or a == b)
// Generic (+/-) pair packing of the last two (equal, size v) indices of in[rows][v][v]:
//   Op[r, P(c,d)] = in[r,c,d] + in[r,d,c] (c >= d),  Om[r, Q(c,d)] = in[r,c,d] - in[r,d,c] (c > d); rows padded with zeros to ldp / ldm
int dev_pack_pm_cols(int64_t rows, int64_t v, const double* in, double* Op, int64_t ldp, double* Om, int64_t ldm);
// out[i,j,:] = Xp[P(i,j),:] + Xm[Q(i,j),:],  out[j,i,:] = Xp[P(i,j),:] - Xm[Q(i,j),:]  (i > j),  out[i,i,:] = Xp[P(i,i),:]
// add (laid out like out): out = scatter + add.  Sp / Sm > 1: Xp / Xm are split-K slabs (stride strideP / strideM apart) that are added up on the way, in slab order
int dev_scatter_pm_rows(int64_t o, int64_t ncols, const double* Xp, const double* Xm, double* out, const double* add = nullptr, int Sp = 1, int64_t strideP = 0,
                        int Sm = 1, int64_t strideM = 0);
// CCSD doubles update, last step in one pass:  t2n[i,j,a,b] = (t2n[i,j,a,b] + OV[i,j,a,b] + U[i,j,a,b] + U[j,i,b,a]) / (eo[i]+eo[j]-ev[a]-ev[b])
int dev_ccsd_finish_t2(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* eo, const double* ev);
// The same with the two ring products taken in place (no transposing accumulation passes over U first) and the t1 denominators on the way:
//   F[i,j,a,b] = U[i,j,a,b] + RS[i,a,j,b] - 1/2 M[i,a,j,b] - M[i,b,j,a]      (RS, M: [o][v][o][v])
//   t2n[i,j,a,b] = t2n[j,i,b,a] = (t2n[i,j,a,b] + OV[i,j,a,b] + F[i,j,a,b] + F[j,i,b,a]) / (eo[i] + eo[j] - ev[a] - ev[b])   -- each (i >= j) pair of tiles once
//   t1n[i,a] /= eo[i] - ev[a]   (t1n may be null)
int dev_ccsd_finish_t2_rings(int64_t o, int64_t v, double* t2n, const double* U, const double* OV, const double* RS, const double* M, const double* eo, const double* ev, double* t1n);
int dev_ladder_scatter_pm(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, double* t2);
// The same with a second pair of packed results added on the way (Hp, Hm; same layouts): p = Rp + f Hp, m = Rm + Hm, where f = 2 on the
// a = b columns and 1 elsewhere -- the hole-hole ladder contracts the packed tau rows, which carry 1/2 on their a = b entries, as its
// RIGHT operand.  assign != 0: t2 = ... instead of t2 += ... (every element of t2 is written).
// Sp / Sm > 1: Rp / Rm are split-K slabs (strideP / strideM apart, rows ldp / ldm long) that are added up on the way, in slab order; Hp / Hm keep
// their own row lengths ldhp / ldhm (0: same as ldp / ldm).
int dev_ladder_scatter_pm2(int64_t o, int64_t v, const double* Rp, int64_t ldp, const double* Rm, int64_t ldm, const double* Hp, const double* Hm,
                           int assign, double* t2, int Sp = 1, int64_t strideP = 0, int Sm = 1, int64_t strideM = 0, int64_t ldhp = 0, int64_t ldhm = 0);
// (+/-) pair-packed images of W[k,l,i,j] (o^4, symmetric under (k,l,i,j) -> (l,k,j,i)) for the hole-hole ladder R[ij,ab] = W[klij] tau[klab]:
//   Ap[P(ij)][P(kl)] = W[klij] + W[klji] (k > l), W[kkij] (k = l), i >= j;   Am[Q(ij)][Q(kl)] = W[klij] - W[klji], k > l, i > j
// (row-major with leading dimensions lda_p >= npair(o), lda_m >= npair'(o); padding columns zeroed)
int dev_pack_w_pm(int64_t o, const double* W, double* Ap, int64_t lda_p, double* Am, int64_t lda_m);
// the same for W[k,l,i,j] = Wt[i,j,k,l] + X[i,j,k,l] + At[j,i,k,l] + At[i,j,l,k] formed on the fly (the Woooo intermediate of the CCSD update: never stored; every
// operand with the row pair (i,j) of the packed images as its slow indices, so that a row of the images reads contiguous blocks)
int dev_pack_w_pm_sum(int64_t o, const double* Wt, const double* X, const double* At, double* Ap, int64_t lda_p, double* Am, int64_t lda_m);
// t1n[i,a] = sum_c t1[i,c] Lvv[a,c] - sum_k Loo[k,i] t1[k,a] + sum_k Q[i,k] t1[k,a],  Q[i,k] = sum_c t1[i,c] Fov[k,c]   (the four small products of the T1 equation)
int dev_ccsd_t1_small(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, double* t1n);
// The whole right-hand side of the T1 equation in one launch, one workgroup per element (i,a):
//   t1n[ia] = [the small products of dev_ccsd_t1_small] + S[(ia),:] . Fov + Lph1[(ia),:] . t1 + sum_s PA[s][(ia)] - sum_s PB[s][(ia)]
// PA / PB: the two long-K products (ovvv . Theta, Lovoo . T) as the split-K slabs their GEMMs left (SA / SB of them, strideA / strideB apart)
int dev_ccsd_t1_assemble(int64_t o, int64_t v, const double* t1, const double* Lvv, const double* Loo, const double* Fov, const double* S, const double* Lph1,
                         const double* PA, int SA, int64_t strideA, const double* PB, int SB, int64_t strideB, double* t1n);
// two independent matrix-vector passes in one launch: y1 = a1 T1 x1 + b1 y1 (rows1 x cols1) and y2 = a2 T2 x2 + b2 y2 (rows2 x cols2)
int dev_gemv_rows_two(int64_t rows1, int64_t cols1, const double* T1, int64_t ld1, const double* x1, double* y1, double a1, double b1,
                      int64_t rows2, int64_t cols2, const double* T2, int64_t ld2, const double* x2, double* y2, double a2, double b2);
// y[r] = alpha (T1[r,:] . x1 + T2[r,:] . x2) + beta y[r]: two matrix-vector products in one pass
int dev_gemv_rows2(int64_t rows, int64_t cols, const double* T1, int64_t ld1, const double* x1, const double* T2, int64_t ld2, const double* x2, double* y, double alpha, double beta);
// F[k,i] = sum_l (2 X[i,l,k,l] - X[l,i,k,l]) for X[i,j,k,l] (o^4): the occupied-occupied intermediate sum_{lcd} (2 ovov[kcld] - ovov[kdlc]) tau[ilcd]
// as a partial trace of X[i,j,k,l] = sum_cd ovov[kcld] tau[ijcd], which the Woooo build forms anyway
int dev_foo_from_x(int64_t o, const double* X, double* F);
// The particle-hole layouts of t2 for the ring terms, all from ONE pass over t2 (t2: [o][o][v][v]; every output [o][v][o][v]):
//   T [k,c,j,b] = t2[k,j,c,b]        Tp[k,c,j,b] = t2[k,j,b,c]        S = 2 T - Tp
//   Ut[k,c,j,b] = S  - 2 t1[j,c] t1[k,b]        Tpt[k,c,j,b] = Tp + 2 t1[j,c] t1[k,b]
// and, in the layout of t2 itself ([o][o][v][v]),  Th[k,j,c,b] = 2 t2[k,j,b,c] - t2[k,j,c,b]
int dev_ccsd_ph_layouts(int64_t o, int64_t v, const double* t2, const double* t1, double* T, double* Tp, double* S, double* Ut, double* Tpt, double* Th);
// Batched small-K update (K = n_occ): C[z][m][n] += alpha sum_k A[z][k][m] B[z][k][n]   (A: [K][M] per batch, stride sA; B: [K][N] per batch,
// stride sB; a stride of 0 shares the operand; C: [M][N] per batch, stride sC).  One pass over C -- these products are HBM bound, a
// tiled GEMM with two k-steps is not.
int dev_small_k_update(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sA, const double* B, int64_t sB,
                       double* C, int64_t sC);
// Y[a,c] = 2 sum_k ZC[k,k,a,c] - sum_k ZB[k,c,a,k]   (ZC: [o][o][v][v], ZB: [o][v][v][o]; the k = i traces of the two ovvv.t1 products)
// add ([v][v], or S split-K slabs of it `stride` apart that are added up in slab order): Y = traces + scale * add
int dev_ccsd_y_traces(int64_t o, int64_t v, const double* ZC, const double* ZB, double* Y, const double* add = nullptr, int S = 1, int64_t stride = 0, double scale = 1.0);

// ---- screening helpers of the semi-sparse DF transform ---------------------------------------------------------------
// out[i] = (|x[i]| >= eps) ? 1 : 0
int dev_threshold_mask(int64_t n, const double* x, double eps, double* out);
// x[r*cols + c] *= m[c]   for r < rows   (broadcast a mask / scale row over a batch of rows)
int dev_mul_bcast_rows(int64_t rows, int64_t cols, double* x, const double* m);

// Absolute overlap int |chi_a| |chi_b| of UNNORMALISED uncontracted Cartesian Gaussians chi = x^i y^j z^k exp(-alpha r^2) by Gauss-Hermite
// quadrature per Cartesian direction (molbe/eri_sparse_DF.py:733-812 `_primitive_overlap`, :815-865 `_primitive_overlap_matrix`): the
// screening matrix of the semi-sparse DF pipeline.  nsh primitive shells with angular momentum l[s] <= 4, exponent ex[s], centre
// xyz[3s..3s+2] and first Cartesian function cart0[s] (components in libcint order, cart_components()); out: ncart x ncart row-major.
int dev_abs_overlap_prim(int nsh, const int* l, const double* ex, const double* xyz, const int64_t* cart0, int64_t ncart, int nroots,
                         const double* roots, const double* weights, double* out);

// dst[r, 0:len] = idx[r] >= 0 ? src[idx[r]*ld + 0:len] : 0   (row gather; idx is an int64 array ON THE DEVICE; dst rows are len long)
int dev_gather_rows(int64_t nrows, int64_t len, const int64_t* idx_dev, const double* src, int64_t ld, double* dst);
// x[r, 0:len] *= s[r]
int dev_scale_rows(int64_t nrows, int64_t len, double* x, const double* s);

// ---- reductions ---------------------------------------------------------------------------------
// out_dev[0] = sum_i x[i]*y[i]   (deterministic two-stage reduction; out_dev is a device double)
int dev_dot(int64_t n, const double* x, const double* y, double* out_dev);
// out_dev[0] = max_i |x[i]|
int dev_absmax(int64_t n, const double* x, double* out_dev);
// out_dev[j] = <x, ys[j]> for j < m <= 8 in one pass over x (each result bit-identical to dev_dot)
int dev_dot_many(int64_t n, const double* x, int m, const double* const* ys, double* out_dev);
// DIIS push for an error vector that is a difference, ONE pass and one launch:  e[i] = trial[i] - prev[i],  xcopy[i] = trial[i] (xcopy may be
// null, and may alias prev),  row[j] = <e, ys[j]> for j < m <= 8 with ys[self] == e (the new vector's own slot).  The row lands in row_dev AND in
// the pinned host block row_host (written by the last workgroup of the launch: no copy node, valid after a wait of this context's stream).
// Same partition and summation order as dev_dot / dev_dot_many.
// flag_host (may be null): a pinned host word that receives `seq` AFTER the results (system-scope release): dev_wait_flag spins on it -- a
// microsecond or two after the kernel's last store instead of the ~13 us of a stream wait.
int dev_diis_push(int64_t n, const double* trial, const double* prev, double* e, double* xcopy, int m, const double* const* ys, int self,
                  double* row_dev, double* row_host, void* flag_host = nullptr, unsigned long long seq = 0);
// Collected launches: between dev_batch_begin and dev_batch_flush the CALLING THREAD's dev_diis_push / dev_ccsd_extrapolate_energy calls are kept, not launched
// (whatever context is bound at the time of the call); the flush issues them -- the same kernel of up to eight calls in one grouped launch -- on the stream of the
// context bound at the flush.  The lock-step sweep ends an iteration of six fragments with four launches instead of twenty-four.
int dev_batch_begin();
int dev_batch_flush();
// wait until the pinned host word holds `seq` (written by a kernel of THIS context's stream; a stream that has drained without writing it is an error)
int dev_wait_flag(const void* flag_host, unsigned long long seq);
// End of a CCSD iteration in one launch:  amp = sum_k coef[k] xs[k] over the packed amplitudes [t1 (o v) | t2 (o,o,v,v)] (nterms <= 8; amp may
// be xs[0] when nterms == 1 and coef[0] == 1: nothing is rewritten),  tau = t2 + t1 (x) t1 of the NEW amplitudes,  E = <L, tau> to e_dev and to
// the pinned host word e_host (last workgroup).  o * o <= 16384.
int dev_ccsd_extrapolate_energy(int64_t o, int64_t v, int nterms, const double* coef, const double* const* xs, double* amp, const double* L,
                                double* tau, double* e_dev, double* e_host, void* flag_host = nullptr, unsigned long long seq = 0);

// ---- matrix-vector style contractions for J/K builds (HBM bound) -------------------------------
// y[r] = alpha * sum_c T[r*ldt + c] * x[c] + beta*y[r]        r < rows, c < cols
int dev_gemv_rows(int64_t rows, int64_t cols, const double* T, int64_t ldt, const double* x,
                  double* y, double alpha, double beta);
// y[r] = alpha * sum_b sum_c T[b*strideT + r*ldt + c] * x[b*stridex + c] + beta*y[r]   (sum over a batch of matrices)
int dev_gemv_rows_batched(int64_t rows, int64_t cols, int64_t nbatch, const double* T, int64_t ldt, int64_t strideT,
                          const double* x, int64_t stridex, double* y, double alpha, double beta);
// Y[p*ldy + r] = alpha * sum_m x[m] * T[(p*mid + m)*inner + r] + beta*Y   p<outer, m<mid, r<inner
int dev_contract_mid(int64_t outer, int64_t mid, int64_t inner, const double* T, const double* x,
                     double* Y, int64_t ldy, double alpha, double beta);

// ---- packed-pair index transforms ------------------------------------------------------------------
// pair index ij = i(i+1)/2 + j, i >= j   (reference: shared/helper.py:260-276 ravel_symmetric)
// s4 (npair x npair)  ->  s1 (n^4, [i,j,k,l])
int dev_unpack_s4(int64_t n, const double* s4, double* s1);
// s1 -> s4 (reads i>=j, k>=l elements)
int dev_pack_s4(int64_t n, const double* s1, double* s4);
// s8 (1-D npair(npair)) -> s4
int dev_unpack_s8_to_s4(int64_t n, const double* s8, double* s4);
// rows of a (rows x npair(n)) packed matrix -> (rows x n x n) full symmetric, and back
int dev_unpack_tril_rows(int64_t rows, int64_t n, const double* packed, double* full);
// the same with a row stride ld >= n of the n x n images (full: rows x n x ld).  A stride that is a multiple of 16 doubles keeps every written
// run on whole 128-byte lines (n = 220: 3.7 -> 5.7 TB/s); the padding columns are not written.
int dev_unpack_tril_rows_ld(int64_t rows, int64_t n, int64_t ld, const double* packed, double* full);
int dev_pack_tril_rows(int64_t rows, int64_t n, const double* full, double* packed);
// full[P(x,y)][k][l] = in[(x*nr + y)][P(k,l)], x >= y (x, y < nr; k, l < n): pair-row selection of an (nr*nr) x npair(n) matrix
// fused with the unpack of its pair column
int dev_unpack_tril_pair_rows(int64_t nr, int64_t n, const double* in, double* full);
int dev_unpack_tril_pair_rows_ld(int64_t nr, int64_t n, int64_t ld, const double* in, double* full);      // full: npair(nr) x n x ld

// ---- symmetric eigen / SVD by wavefront Jacobi (no MFMA) -----------------------------------------
// A (n x n, symmetric, row-major, overwritten) -> eigenvalues w[n] ascending and eigenvectors in the
// COLUMNS of V (n x n row-major).  sweeps_out (host int*) may be null.
int dev_jacobi_eigh(int64_t n, double* A, double* w, double* V, int* sweeps_out);
// The same with an early stop: the sweeps end once the largest normalised off-diagonal element met during a sweep is below `stop_below`
// (what is left after that sweep is its square).  dev_jacobi_eigh uses 1e-10 -- nothing above the rotation threshold is left; a caller
// that will diagonalise again anyway (the cycles of an SCF before the last) can stop a sweep earlier.
int dev_jacobi_eigh_until(int64_t n, double* A, double* w, double* V, int* sweeps_out, double stop_below);
// One-sided Jacobi SVD of G (m x n row-major, m >= n), overwritten by U*diag(s) columns;
// s[n] descending, V (n x n) right vectors in columns, U (m x n) left vectors in columns.
// ---- fused steps of the fragment RHF of SMALL fragments (n <= dev_scf_fused_max(); 0: not available / switched off): see linalg_f64.hip
int dev_scf_fused_max();
// eigenproblem of F in the basis Cp (nullptr: as given): w ascending, C_out = Cp V (columns in that order), the same to C2_out (nullable; may be Cp), dm_out (nullable) =
// 2 C_occ C_occ^T of the lowest nocc columns.  Asynchronous: *status_dev (device) receives the number of sweeps, or -1 when 40 sweeps did not converge -- read it at the
// caller's next transfer.
int dev_jacobi_eigh_in_basis(int64_t n, const double* F, const double* Cp, double* w, double* C_out, double* C2_out, int nocc, double* dm_out, double stop_below,
                             int* status_dev);
// F = h + J - K/2, err = F D - D F, scal2[0] = sum (h + F) o D, scal2[1] = sum err^2 (device)
int dev_scf_fock_small(int64_t n, const double* h, const double* J, const double* K, const double* D, double* F, double* err, double* scal2);
// Dp[P(r,s)] = D[r,s] + D[s,r] (r > s), D[r,r]
int dev_pack_density_sym(int64_t n, const double* D, double* Dp);
int dev_jacobi_svd(int64_t m, int64_t n, double* G, double* s, double* U, double* V, int* sweeps_out);

// ---- the one exchange of the sharded sweep: a persistent communicator, one process per GPU ----------------
// The reference has no communication backend: be_func_parallel pickles whole result tuples back through pathos pipes
// (molbe/be_parallel.py:484-517).  Here each rank drives one GPU and the only exchange per sweep is an all-reduce of a few kB.
// Product: RCCL (ncclAllReduce, ncclDouble) on the default context's stream, communicator created once (comm_rccl.hip).
// dev_comm_unique_id: 128 opaque bytes made by ONE rank and handed to every rank out of band (file, socket, MPI, ...).
enum { COMM_ID_BYTES = 128, COMM_SUM = 0, COMM_MAX = 1 };
int dev_comm_unique_id(void* id128);
int dev_comm_init(int rank, int world, const void* id128);
int dev_comm_info(int* rank, int* world);                    // 0 / 1 without a communicator
// in place on a HOST buffer of n doubles; every rank gets the identical result
int dev_comm_allreduce(double* host_buf, int64_t n, int op);
int dev_comm_destroy();

// ---- Cholesky / triangular inverse (DF metric) ---------------------------------------------------------
// A (n x n SPD row-major) -> L lower triangular with A = L L^T (upper part zeroed), in place
int dev_cholesky_lower(int64_t n, double* A);
// Linv = L^{-1} (lower triangular, row-major n x n)
int dev_tri_inverse_lower(int64_t n, const double* L, double* Linv);

}  // namespace qemb
