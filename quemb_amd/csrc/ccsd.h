// ccsd.h -- device-resident RCCSD for one embedded fragment (row a8 of SURVEY.md section 8) plus the
// embedding->MO integral transformation that feeds it and the contracted RDM/energy pieces (a9-a11).
#pragma once
#include <cstdint>
#include <vector>
#include "dev_ops.h"
#include "tensor_utils.h"

namespace qemb {

struct CcsdOptions {
  double conv_tol = 1e-10;        // |dE|      (PySCF default 1e-7; tighter here)
  double conv_tol_normt = 1e-8;   // |dt|      (PySCF default 1e-5)
  int max_cycle = 100;            //           (PySCF default 50)
  int diis_space = 6;
  int verbose = 0;
};

// MO-basis integral blocks, chemists' notation, all device resident and contiguous.
struct MoIntegrals {
  int n = 0, o = 0, v = 0, nf = 0;
  DBuf oooo, ovoo, ovov, oovv, ovvo, ovvv;
  DBuf Vl;          // Vl[a,b,c,d] = (ac|bd)   (dense ladder operand; only built on request: export / measurement)
  // (+/-) packed ladder operands: Vp[P(a,b),P(c,d)] = (ac|bd)+(ad|bc), Vm[Q(a,b),Q(c,d)] = (ac|bd)-(ad|bc)
  DBuf Vp, Vm;
  int64_t ldp = 0, ldm = 0;
  DBuf A1, A2;      // A1[a,j,b,P] = (P a|j b), A2[i,j,b,P] = (P i|j b), P < nf in the EMBEDDING basis
  DBuf T34;         // T34[q',r',s',P] = (P q'|r' s'), P < nf: all 3/4-transformed integrals (relaxed-density energy only)
};

// eri_s4: (npair x npair) 4-fold packed embedding-basis ERIs on the device (read only); X0, X1: two device work
// buffers of mo_transform_work(n) = n^2 * npair doubles each; C: n x n MO coefficients (columns) on the device.
// x1_is_unpacked: X1 already holds the half-unpacked tensor [P(p,q)][r][s] (it is consumed).
int mo_slab_ld(int n);            // row stride of the unpacked n x n images inside X0 / X1 (>= n)
int64_t mo_transform_work(int n);
int mo_transform(int n, int o, int nf, const double* eri_s4, double* X0, double* X1, const double* C, MoIntegrals& out,
                 bool build_Vl = false, bool build_T34 = false, bool x1_is_unpacked = false);
// the same blocks from the fragment's 3-index factor B[naux][npair(n)] (bb of molbe/eri_onthefly.py:141; eri = bb^T bb, :143): the factor is
// transformed and multiplied with itself in the MO basis -- 2 naux npair^2 flops instead of 8 n^3 npair (ccsd.cpp)
int mo_transform_factor(int n, int o, int nf, int naux, const double* B_packed, double* X0, double* X1, const double* C, MoIntegrals& out,
                        bool build_Vl = false, bool build_T34 = false);
// which route a solve takes when the fragment holds a factor: the factor's, while its product is the cheaper one (n = 220: 4.8 ms + 9.6 ms per 660 auxiliary
// functions against 41 ms for the four quarter transformations -- the times cross near naux = 11 n; 8 n leaves a margin)
inline bool mo_factor_route_pays(int n, int naux) { return naux > 0 && (int64_t)naux <= 8 * (int64_t)n; }

class CcLambda;

struct LockstepStats { long long merged_runs = 0, launches = 0, grouped = 0, operations = 0, max_group = 0; double ms_tapes = 0.0, ms_post = 0.0; };
class CcsdSolver;
int ccsd_kernel_lockstep(const std::vector<CcsdSolver*>& s, const std::vector<CcsdOptions>& opt, const std::vector<int>& ctx, int home_ctx,
                         std::vector<double>& e_corr, std::vector<int>& n_iter, std::vector<char>& converged, LockstepStats* stats);

class CcsdSolver {
  friend class CcLambda;
  friend int ccsd_kernel_lockstep(const std::vector<CcsdSolver*>&, const std::vector<CcsdOptions>&, const std::vector<int>&, int,
                                  std::vector<double>&, std::vector<int>&, std::vector<char>&, LockstepStats*);
 public:
  int setup(MoIntegrals&& ints, const double* mo_energy_dev);
  // defer_energy: the energy of the guess stays on the device (the launches may be captured for a tape); fetch_energy() reads it back afterwards
  int init_amps(bool defer_energy = false);         // MP2 guess (t1 = 0 for the diagonal Fock)
  int set_amps(const double* t1_dev, const double* t2_dev, bool defer_energy = false);   // warm start
  int fetch_energy();
  int iterate(double* e_corr, double* normt);       // one update_amps + DIIS + energy
  int iterate_update(bool prefer_tape, bool defer_tape, bool* deferred);   // the two halves of iterate(), for the lock-step sweep
  int iterate_post(double* e_corr, double* normt);
  bool fused_post() const;
  int post_wait(int step);
  int post_issue();                                 // iterate_post in three steps (a wait of this context's stream between them)
  int post_extrapolate(double* normt);
  int post_energy(double* e_corr);
  int prepare_tape(int peers = 1);                               // record the update as a tape before the first iteration (lock-step sweeps)
  int kernel(const CcsdOptions& opt, double* e_corr, int* n_iter, bool* converged);
  // energy pieces of get_frag_energy that need t1,t2: Z1[i,P], Z2[a,P] (host outputs o*nf and v*nf)
  int energy_intermediates(std::vector<double>& Z1, std::vector<double>& Z2);
  // copy a named integral block to the host (measurement / debugging): oooo ovoo ovov ovvv Vl W1base W2base eo ev
  int export_block(const char* name, double* host, int64_t nelem);
  // out[i,j,a,b] += sum_cd (ac|bd) x[i,j,c,d] through the (+/-) pair-packed operands; x must satisfy x[j,i,d,c] = x[i,j,c,d]
  // rows_packed: LTp_/LTm_ already hold the packed rows of x.  hh: also add the hole-hole ladder Woooo[klij] x[klab] from the packed images
  // WAp_/WAm_ of Woooo, and ASSIGN the sum to out (first writer) instead of accumulating
  int apply_ladder(const double* x, double* out, bool rows_packed = false, bool hh = false);
  const MoIntegrals& integrals() const { return I_; }
  double* t1() { return amp_.p; }
  double* t2() { return amp_.p + (int64_t)o_ * v_; }
  int o() const { return o_; }
  int v() const { return v_; }
  int64_t n_amp() const { return (int64_t)o_ * v_ + (int64_t)o_ * o_ * v_ * v_; }

 private:
  int update_amps(double* t1n, double* t2n);
  int energy(const double* t1, const double* t2, double* e);
  int make_tau(const double* t1, const double* t2, double* tau);

  int o_ = 0, v_ = 0, nf_ = 0;
  MoIntegrals I_;
  DBuf eo_, ev_;
  // derived constant tensors
  DBuf ovov_t_, Lovov_, Loovv_, OVoovv_, Lovoo_, ovoo_ijka_, ovoo_kilc_, W1base_, W2base_, Lph1_, OVp_, OVm_, oooo_p_, ovoo_cikl_;   // OVp/OVm: (+/-) pair-packed OVl[k,a,c,d] = ovvv[k,d,a,c] over (c,d)
  // amplitudes (t1 then t2, one contiguous vector) and per-iteration work space
  DBuf amp_, ampn_, diff_;
  DBuf tau_, T_, Tp_, S_, W1_, W2_, W12_, W12b_, R_, U_, G2_, T1P_;      // T1P_: slabs of the two long-K products of the T1 equation
  DBuf LTp_, LTm_, LRp_, LRm_;   // (+/-) packed ladder: tau combinations and results
  DBuf Xp_, Xm_;                 // (+/-) packed rows of X[i,j,k,a] = tau[ijcd] ovvv[kdac]
  DBuf ovvv_pk_, ZCp_;           // ovvv[k,d,(a>=c)] packed over its symmetric pair, and the packed result of its t1 contraction
  DBuf Gp_, Gm_;                 // (+/-) pair-packed images over (c,d) of ovov[k,c,l,d], one row per (k,l): the Woooo build contracts packed tau rows
  DBuf Xwp_, Xwm_, Xw_;          // its (+/-) packed result rows [P(ij)][(kl)] and their expansion [i,j,k,l]
  DBuf WAp_, WAm_, HRp_, HRm_;   // hole-hole ladder: (+/-) packed images of Woooo and its packed result rows
  int64_t lwp_ = 0, lwm_ = 0;    // leading dimensions of WAp_ / WAm_
  DBuf ZB_, ZC_;                 // ZB[k,c,a,i] = ovvv[kcad] t1[id],  ZC[k,i,a,c] = t1[id] ovvv[kdac]  (one ovvv pass each per iteration)
  DBuf Foo_, Fvv_, Fov_, Z_, Y_, Ytmp_, Loo_, Lvv_, Q_, Wo_, O1_, X_, scal_, LovooT_;
  std::vector<DeviceDIIS> diis_;
  bool first_ = true;
  // hipGraph of one update_amps (small fragments are launch bound: ~170 launches of a few microseconds each)
  dev_graph_t graph_ = nullptr;
  dev_tape_t tape_ = nullptr;       // the same launch sequence as data (dev_tape_end): executed together with other fragments' tapes
  int eager_iters_ = 0;
  bool graph_ok_ = true;
  double* last_out_ = nullptr;      // where iterate_update put the new amplitudes; iterate_post continues from there
  double* host_scal_ = nullptr;     // pinned: [energy, |dt|^2] on their way back, then the words the fused launches publish behind them (energy step, push step)
  unsigned long long seq_push_ = 0, seq_energy_ = 0;      // ... and their expected values
  bool post_pending_ = false;                             // a fused launch has been issued and not yet waited for
  bool last_use_diis_ = false, last_replayable_ = false;
 public:
  // (a launch that will still write its result and sequence word into the pinned block -- a solve that left through an error between issue and wait -- must have
  //  finished before the block is parked for the next solver: wait for the device in that case only)
  ~CcsdSolver() {
    if (graph_) dev_graph_destroy(graph_);
    if (tape_) dev_tape_destroy(tape_);
    if (host_scal_) { if (post_pending_) (void)dev_sync_device(); dev_pinned_free(host_scal_); }
  }
  CcsdSolver() = default;
  // the recorded update of a lock-step sweep changes hands: a fragment keeps it from one solve to the next when the next solver's buffers are where this one's
  // were (Fragment::tape_for_lockstep)
  dev_tape_t release_tape() { dev_tape_t t = tape_; tape_ = nullptr; return t; }
  void adopt_tape(dev_tape_t t) { if (tape_) dev_tape_destroy(tape_); tape_ = t; }
  dev_tape_t tape() const { return tape_; }
  CcsdSolver(const CcsdSolver&) = delete;
  CcsdSolver& operator=(const CcsdSolver&) = delete;
 private:
  double ecc_ = 0.0;
};

// tile configuration and split-K factor CcsdSolver picks for a "few packed pair rows x many columns" product (introspection for tests / tools)
void pick_pair_gemm(int64_t rows, int64_t cols, int& cfg, int& ks, int64_t K = 0);
}  // namespace qemb
