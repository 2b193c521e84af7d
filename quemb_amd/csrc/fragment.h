// fragment.h -- one embedded fragment resident on the device and the body of the fragment sweep
// (be_func's loop body, molbe/solver.py:301-547 == run_solver, molbe/be_parallel.py:40-307).
#pragma once
#include <string>
#include <cstdint>
#include <memory>
#include <vector>
#include "ccsd.h"
#include "cc_lambda.h"
#include "scf.h"

namespace qemb {

struct FragmentOptions {
  CcsdOptions cc;
  ScfOptions scf;
  int warm_start = 0;      // reuse converged t1/t2 of the previous solve as the CCSD guess (reference restarts from MP2)
  int keep_amplitudes = 1; // keep t1/t2 resident after the solve
  int relax_density = 0;   // solve_ccsd(relax=True), solver.py:925-939: Lambda equations + response densities
  LambdaOptions lam;
  int strict = 1;          // 0: non-convergence is reported (QEMB_WARN_NOCONV) instead of being an error, like PySCF's warnings
};

struct FragmentResult {
  int n_iter = 0;
  int lambda_iters = 0;
  int scf_cycles = 0;
  int ccsd_converged = 0, scf_converged = 0;
  double e_corr_mo = 0.0;      // CCSD correlation energy of the embedding problem
  double e_scf = 0.0;          // fragment RHF energy (electronic, of h = fock + heff)
  double e_frag[3] = {0, 0, 0};   // [e1, e2, ec] weighted centre sums (get_frag_energy, helper.py:333-339)
  double ebe_hf = 0.0;         // update_ebe_hf (pfrag.py:327-400) evaluated with the SCF orbitals of this solve
};

class Fragment {
 public:
  Fragment(int n, int nf) : n_(n), nf_(nf) {}
  int last_lambda_iters = 0;
  int n() const { return n_; }
  int nf() const { return nf_; }
  int o() const { return o_; }
  // ERIs: 4-fold packed (npair x npair), the layout of dataset "f{I}" (mbe.py:1039)
  int set_eri_s4_host(const double* s4);
  int set_eri_s4_dev(const double* s4_dev);     // device-to-device copy
  int adopt_eri_s4(DBuf&& s4);                  // takes the block a transform just produced (no copy: 4.7 GB at n = 220)
  double* eri_s4() { return eri_s4_.p; }
  // The fragment's 3-index factor B[naux][npair(n)] with eri_s4 = B^T B (bb of molbe/eri_onthefly.py:141-143), optional.  With it the MO integrals
  // of a solve come from the factor (mo_transform_factor, ccsd.cpp) while that is the cheaper route; set it AFTER the ERIs (new ERIs drop it).
  int set_df_factor_host(int naux, const double* B);
  int set_df_factor_dev(int naux, const double* B_dev);
  int adopt_df_factor(DBuf&& B, int naux);
  void clear_df_factor();
  // A fragment that LIVES on its factor (round 5): no 4-fold packed block is resident -- 8 naux npair bytes instead of 8 npair^2 (128 MB instead of 4.7 GB at
  // n = 220, naux = 660).  J / K of the fragment RHF, the MO integrals, the energies and the CPHF response all come from the factor; the block is formed on
  // demand only (export_eri_s4, or a solve with the four-index route forced: a transient of that solve).  Any set_eri_s4 / adopt_eri_s4 ends the mode.
  int set_df_only_host(int naux, const double* B);
  int set_df_only_dev(int naux, const double* B_dev);
  int adopt_df_only(DBuf&& B, int naux);
  bool factor_only() const { return !eri_s4_.p && df_factor_.p; }
  bool has_eris() const { return eri_s4_.p || df_factor_.p; }
  int export_eri_s4(double* s4_host);               // the resident block, or B^T B formed for this call
  int64_t resident_bytes() const;                   // device bytes this fragment keeps between solves (ERIs / factor, orbitals, kept amplitudes)
  int df_naux() const { return df_naux_; }
  int set_mo_route(int route);                  // -1: by cost (default), 0: four-index transformation of eri_s4, 1: the factor (an error without one)
  bool use_factor_route() const;
  bool last_route_was_factor() const { return last_route_factor_; }
  // static data for the energies: h1, veff0 (n x n host), centre weight/indices
  void set_energy_data(const double* h1, const double* veff0, const double* veff, double weight, const int* centers, int ncen);
  // The sweep body.  h = fock + heff (n x n host), dm0 (n x n host, may be null -> core guess).
  // Outputs (host, nullable): mo_coeff n*n, mo_energy n, rdm1_emb n*n (= C rdm1 C^T / 2), rdm1_mo n*n, t1 o*v, t2 o*o*v*v.
  int solve(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res,
            double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1, double* t2);
  // The same in three steps -- everything before the CCSD iterations, (the iterations: cc_->kernel or the lock-step loop of solve_batch),
  // everything after -- so that several fragments can share the middle step.
  int solve_begin(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res);
  // ... and solve_begin itself in two halves: the fragment RHF (host round trips inside), then MO integrals + CCSD set-up + starting amplitudes -- a pure launch
  // sequence, which solve_batch records as a tape per fragment and runs merged for all fragments (defer_energy: the guess's energy is fetched afterwards)
  int solve_begin_scf(int o, const double* h, const double* dm0, const FragmentOptions& opt, int eeval, FragmentResult* res);
  int solve_begin_cc(bool defer_energy);
  int solve_end(double* mo_coeff, double* mo_energy, double* rdm1_emb, double* rdm1_mo, double* t1, double* t2);
  struct BatchOutputs { double *mo_coeff = nullptr, *mo_energy = nullptr, *rdm1_emb = nullptr, *rdm1_mo = nullptr, *t1 = nullptr, *t2 = nullptr; };
  // every fragment of a sweep in one call: solve_begin / solve_end per fragment on its own execution context (host thread + stream),
  // the CCSD iterations of all of them in lock step (one grouped launch per operation).  Results as from solve(), bit for bit.
  static int solve_batch(const std::vector<Fragment*>& frs, const std::vector<int>& o, const std::vector<const double*>& h,
                         const std::vector<const double*>& dm0, const FragmentOptions& opt, int eeval, std::vector<FragmentResult>& res,
                         const std::vector<BatchOutputs>& outs, LockstepStats* stats);
  // Bench hooks: set up the CCSD problem once (SCF + transform), then time single iterations.
  int prepare_ccsd(int o, const double* h, const double* dm0, const FragmentOptions& opt);
  int ccsd_iterate(int niter, double* e_corr, double* normt);
  int ccsd_export(const char* name, double* host, int64_t nelem) {
    if (!cc_) { set_error("Fragment: prepare_ccsd first"); return QEMB_ERR_ARG; }
    if (name && std::string(name) == "mo_coeff") {      // the orbitals the MO integrals were transformed with
      if (nelem != (int64_t)n_ * n_) { set_error("export_block: element count mismatch"); return QEMB_ERR_ARG; }
      return dev_d2h(host, C_, sizeof(double) * nelem);
    }
    return cc_->export_block(name, host, nelem);
  }
  int ccsd_reset();                              // back to the MP2 guess
  // fragment RHF only (Frags.scf(fs=True), mbe.py:1160): outputs host n*n / n / n*n / n*n, nullable
  int scf_only(int o, const double* h, const double* dm0, const ScfOptions& opt, double* mo_coeff, double* mo_energy,
               double* J_host, double* K_host, ScfResult* sres);
  // CPHF response of the fragment RHF density to npot one-body perturbations v_p (n x n each, host):
  // dP_p = d(P)/d(lambda_p), P the spin-summed... see fragment.cpp.  dPs: npot x n x n (host).
  int cphf_response(int o, const double* h, const double* dm0, const ScfOptions& opt, const double* vpots, int npot, double* dPs);
  // J/K based pieces that do not need a correlated solve (row a6 / a15)
  int hf_veff_from_dm(const double* P_host, double* J_host, double* K_host);   // J,K of an n x n density

 private:
  int run_scf(int o, const double* h, const double* dm0, const ScfOptions& opt, double* X0, ScfResult* sres, bool warm = false);
  int materialize_s4(DBuf& out);                    // B^T B over packed pairs
  const double* s4_ptr() const { return eri_s4_.p ? eri_s4_.p : s4_transient_.p; }
  DBuf s4_transient_;                               // factor-only fragment, four-index route forced: the block for the duration of one solve
  int scf_operand(DBuf& X1, bool* unpacked);
  int check_df_factor();
  int mo_integrals(int o, int nf, DBuf& X1, bool x1_unpacked, MoIntegrals& ints, bool build_Vl, bool build_T34);
  DBuf df_factor_; int df_naux_ = 0; int mo_route_ = -1; bool last_route_factor_ = false;
  bool have_C_ = false; int c_nocc_ = -1;    // C_ holds the orbitals of a converged earlier solve with c_nocc_ occupied orbitals (the Jacobi eigensolver starts in that basis)
  int n_, nf_, o_ = -1;
  DBuf eri_s4_;
  std::vector<double> h1_, veff0_, veff_;
  double weight_ = 1.0;
  std::vector<int> centers_;
  // what solve_begin hands to solve_end
  struct SolvePending {
    int o = 0, eeval = 0; FragmentOptions opt; FragmentResult* res = nullptr;
    bool unconverged = false, no_virtuals = false;
    std::vector<double> C, eps;
    DBuf X1; bool x1_unpacked = false;      // the half-unpacked tensor between the two halves of solve_begin (four-index route)
    unsigned long long alloc_hash = 0;      // dev_alloc_trace_end over integrals + set-up + starting amplitudes of this solve
  } sp_;
  // the recorded amplitude update of the last lock-step solve and the buffer layout it is valid for (tape_for_lockstep / retire_solver)
  struct TapeCache { dev_tape_t tape = nullptr; unsigned long long key = 0; ~TapeCache() { if (tape) (void)dev_tape_destroy(tape); } } tape_cache_;
  unsigned long long tape_key_ = 0;
  int tape_for_lockstep(int peers);
  void retire_solver();
  // state of the last solve
  DBuf C_, eps_, dm_, J_, K_;
  std::unique_ptr<CcsdSolver> cc_;
  DBuf t_prev_;   // warm-start amplitudes (t1 then t2)
  DBuf z_prev_;   // warm-start Lambda multipliers (relax_density)
  int z_prev_o_ = -1;
  int t_prev_o_ = -1;
};

// lock-step tapes kept from an earlier solve / recorded anew since the last reset (all fragments of the process)
void tape_cache_counters(long long* reused, long long* recorded, int reset);

}  // namespace qemb
