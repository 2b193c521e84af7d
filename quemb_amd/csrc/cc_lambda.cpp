// cc_lambda.cpp -- executor of the generated contraction program (see cc_lambda.h).
#include "cc_lambda.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace qemb {

struct CcTensorDef { const char* name; const char* sig; int kind; int zero; };
enum { CC_ES = 0, CC_PERM = 1, CC_LADDER = 2 };
struct CcStmt { int op; int dst; int a; int b; double coef; const char* subs; int flag; };
enum { K_T = 0, K_INT = 1, K_ZERO = 2, K_FWD = 3, K_BAR = 4, K_VIRTUAL = 5 };

#include "cc_lambda_program.inc"

static const int kNumTensors = (int)(sizeof(kCcTensors) / sizeof(kCcTensors[0]));
static const int kNumForward = (int)(sizeof(kCcForward) / sizeof(kCcForward[0]));
static const int kNumBackward = (int)(sizeof(kCcBackward) / sizeof(kCcBackward[0]));

int CcLambda::id_of(const char* name) const {
  for (int k = 0; k < kNumTensors; ++k) if (std::strcmp(kCcTensors[k].name, name) == 0) return k;
  return -1;
}
int64_t CcLambda::dim_of(char c) const { return (c >= 'i' && c <= 'n') ? o_ : v_; }
int64_t CcLambda::numel(const std::string& idx) const { int64_t n = 1; for (char c : idx) n *= dim_of(c); return n; }
int64_t CcLambda::size_of(int id) const {
  int64_t n = 1;
  for (const char* c = kCcTensors[id].sig; *c; ++c) n *= (*c == 'o') ? o_ : v_;
  return n;
}
int CcLambda::settle(int id) {
  if (ptr_[id] && fresh_[id]) { fresh_[id] = 0; return dev_fill(ptr_[id], size_of(id), 0.0); }
  return 0;
}

int CcLambda::ensure(int id) {
  if (ptr_[id]) return 0;
  const int kind = kCcTensors[id].kind;
  if (kind != K_FWD && kind != K_BAR) { set_error("cc_lambda: tensor has no storage"); return QEMB_ERR_ARG; }
  QTRY(buf_[id].alloc(size_of(id)));
  ptr_[id] = buf_[id].p;
  return dev_fill(ptr_[id], size_of(id), 0.0);
}

// dst[didx] = alpha * src[sidx] + beta * dst   (same letters, at most four; shorter tensors are padded with unit dims)
int CcLambda::perm_acc(double* dst, const std::string& didx, const double* src, const std::string& sidx, double alpha, double beta) {
  const int m = (int)sidx.size();
  if (m > 4 || (int)didx.size() != m) { set_error("cc_lambda: permutation of more than four indices"); return QEMB_ERR_ARG; }
  int64_t d[4] = {1, 1, 1, 1};
  int p[4] = {0, 1, 2, 3};
  const int pad = 4 - m;
  for (int k = 0; k < m; ++k) d[pad + k] = dim_of(sidx[k]);
  for (int k = 0; k < m; ++k) {
    const size_t pos = sidx.find(didx[k]);
    if (pos == std::string::npos) { set_error("cc_lambda: index mismatch in permutation"); return QEMB_ERR_ARG; }
    p[pad + k] = pad + (int)pos;
  }
  return perm4(dst, src, d[0], d[1], d[2], d[3], p[0], p[1], p[2], p[3], alpha, beta);
}

// the tensor `id` (stored with index order `have`) in index order `want`: in place, from the cache of constant operands,
// or permuted into `scratch`
int CcLambda::operand(int id, const std::string& have, const std::string& want, const double** out, DBuf& scratch) {
  if (have == want) { *out = ptr_[id]; return 0; }
  if (kCcTensors[id].kind != K_BAR) {
    // position pattern, independent of the letters used by the statement
    std::string key = std::string(kCcTensors[id].name) + ":";
    for (char c : want) key += (char)('0' + have.find(c));
    auto it = cache_.find(key);
    if (it == cache_.end()) {
      DBuf b;
      QTRY(b.alloc(size_of(id)));
      QTRY(perm_acc(b, want, ptr_[id], have, 1.0, 0.0));
      it = cache_.emplace(key, std::move(b)).first;
    }
    *out = it->second.p;
    return 0;
  }
  QTRY(scratch.alloc(size_of(id)));
  QTRY(perm_acc(scratch, want, ptr_[id], have, 1.0, 0.0));
  *out = scratch.p;
  return 0;
}

// dst[so] += coef * sum_K a[sa] b[sb]   as  permute - GEMM - permute, skipping every permutation the GEMM's own
// operand-order flags can absorb
int CcLambda::contract(int dst, double coef, std::string sa, std::string sb, const std::string& so, int a, int b) {
  if (!so.empty() && sa.find(so[0]) == std::string::npos) { std::swap(sa, sb); std::swap(a, b); }
  auto in = [](const std::string& s, char c) { return s.find(c) != std::string::npos; };
  std::string Ma, Nb, Ka, Kb;
  for (char c : sa) { if (in(sb, c) && !in(so, c)) Ka += c; else if (in(so, c) && !in(sb, c)) Ma += c; else { set_error("cc_lambda: unsupported contraction pattern"); return QEMB_ERR_ARG; } }
  for (char c : sb) { if (in(sa, c) && !in(so, c)) Kb += c; else if (in(so, c) && !in(sa, c)) Nb += c; else { set_error("cc_lambda: unsupported contraction pattern"); return QEMB_ERR_ARG; } }
  if (Ma.size() + Nb.size() != so.size() || Ka.size() != Kb.size()) { set_error("cc_lambda: malformed contraction"); return QEMB_ERR_ARG; }
  std::string Mo, No;                       // M and N letters in the order of the output
  for (char c : so) { if (in(Ma, c)) Mo += c; else No += c; }
  const int64_t sizeA = numel(sa), sizeB = numel(sb);
  auto fitsA = [&](const std::string& M, const std::string& K) { return sa == M + K || sa == K + M; };
  auto fitsB = [&](const std::string& K, const std::string& N) { return sb == K + N || sb == N + K; };
  // choose the K order that leaves the larger operand in place
  std::string Kord = Ka;
  {
    const int64_t costA = (fitsA(Ma, Ka) ? 0 : sizeA) + (fitsB(Ka, Nb) ? 0 : sizeB);
    const int64_t costB = (fitsA(Ma, Kb) ? 0 : sizeA) + (fitsB(Kb, Nb) ? 0 : sizeB);
    if (costB < costA) Kord = Kb;
  }
  // an operand that has to be permuted anyway is permuted into the output's index order
  std::string Mord = fitsA(Ma, Kord) ? Ma : Mo;
  std::string Nord = fitsB(Kord, Nb) ? Nb : No;
  const int64_t M = numel(Mord), N = numel(Nord), K = numel(Kord);
  DBuf sA, sB, sC;
  const double *A = nullptr, *B = nullptr;
  bool a_kc = true, b_kc = false;
  if (sa == Mord + Kord) { A = ptr_[a]; a_kc = true; }
  else if (sa == Kord + Mord) { A = ptr_[a]; a_kc = false; }
  else { QTRY(operand(a, sa, Mord + Kord, &A, sA)); a_kc = true; }
  if (sb == Kord + Nord) { B = ptr_[b]; b_kc = false; }
  else if (sb == Nord + Kord) { B = ptr_[b]; b_kc = true; }
  else { QTRY(operand(b, sb, Kord + Nord, &B, sB)); b_kc = false; }
  const int64_t lda = a_kc ? K : M, ldb = b_kc ? K : N;
  static const bool trace = std::getenv("QEMB_LAMBDA_TRACE") != nullptr;
  if (trace) std::fprintf(stderr, "[qemb lambda] %s,%s->%s  M=%lld N=%lld K=%lld a_kc=%d b_kc=%d permA=%d permB=%d permC=%d\n", sa.c_str(), sb.c_str(), so.c_str(),
                          (long long)M, (long long)N, (long long)K, (int)a_kc, (int)b_kc, (int)(sA.p != nullptr), (int)(sB.p != nullptr), (int)(so != Mord + Nord));
  // products with an n_occ-sized side: 32 x 128 / 128 x 32 tiles instead of padding that side to 64 (these are HBM-bound passes
  // over ovvv-sized operands; ccsd.cpp uses the same tiles for the t1 contractions)
  const int cfg = (M <= 32 && N >= 64) ? 21 : (N <= 32 && M >= 64) ? 20 : -1;
  if (so == Mord + Nord) return gemm(M, N, K, coef, A, lda, a_kc, B, ldb, b_kc, take_beta(dst), ptr_[dst], N, 1, 0, 0, 0, cfg);
  QTRY(sC.alloc(M * N));
  QTRY(gemm(M, N, K, 1.0, A, lda, a_kc, B, ldb, b_kc, 0.0, sC, N, 1, 0, 0, 0, cfg));
  return perm_acc(ptr_[dst], so, sC, Mord + Nord, coef, take_beta(dst));
}

int CcLambda::run(const CcStmt& s) {
  QTRY(ensure(s.dst));
  if (s.op == CC_LADDER) {
    if (s.coef != 1.0) { set_error("cc_lambda: ladder statements carry unit coefficients"); return QEMB_ERR_ARG; }
    QTRY(settle(s.dst));
    return cc_.apply_ladder(ptr_[s.a], ptr_[s.dst]);
  }
  const std::string subs(s.subs);
  const size_t arrow = subs.find("->");
  const std::string lhs = subs.substr(0, arrow), so = subs.substr(arrow + 2);
  if (s.op == CC_PERM) return perm_acc(ptr_[s.dst], so, ptr_[s.a], lhs, s.coef, take_beta(s.dst));
  const size_t comma = lhs.find(',');
  return contract(s.dst, s.coef, lhs.substr(0, comma), lhs.substr(comma + 1), so, s.a, s.b);
}

int CcLambda::setup() {
  o_ = cc_.o_; v_ = cc_.v_;
  const int64_t o = o_, v = v_, N2 = o * o * v * v;
  buf_.clear(); buf_.resize(kNumTensors);
  ptr_.assign(kNumTensors, nullptr);
  fresh_.assign(kNumTensors, 0);
  cache_.clear();
  struct { const char* name; double* p; } ext[] = {
      {"t1", cc_.t1()}, {"t2", cc_.t2()}, {"oooo", cc_.I_.oooo.p}, {"ovoo", cc_.I_.ovoo.p}, {"ovov", cc_.I_.ovov.p},
      {"oovv", cc_.I_.oovv.p}, {"ovvo", cc_.I_.ovvo.p}, {"ovvv", cc_.I_.ovvv.p}};
  for (auto& e : ext) {
    const int id = id_of(e.name);
    if (id < 0 || !e.p) { set_error("cc_lambda: missing input tensor"); return QEMB_ERR_ARG; }
    ptr_[id] = e.p;
  }
  for (int k = 0; k < kNumForward; ++k) {
    const CcStmt& s = kCcForward[k];
    if (s.flag) continue;                  // an operand is identically zero (fock = diag(mo_energy))
    QTRY(run(s));
  }
  const int64_t na = o * v + N2;
  QTRY(z_.alloc(na)); QTRY(zn_.alloc(na)); QTRY(diff_.alloc(na)); QTRY(scal_.alloc(4));
  return dev_fill(z_, na, 0.0);
}

int CcLambda::backward(bool lambda_only) {
  const int64_t nov = (int64_t)o_ * v_;
  // adjoints start from zero: allocated buffers are only MARKED (their first writer stores with beta = 0; settle() clears the ones that
  // are read or accumulated into before any write)
  for (int k = 0; k < kNumTensors; ++k)
    if (kCcTensors[k].kind == K_BAR && ptr_[k]) fresh_[k] = 1;
  const int n1b = id_of("n1_bar"), n2b = id_of("n2_bar");
  QTRY(ensure(n1b)); QTRY(ensure(n2b));
  QTRY(dcopy(nov, z_, ptr_[n1b])); fresh_[n1b] = 0;
  QTRY(dcopy(size_of(n2b), z_.p + nov, ptr_[n2b])); fresh_[n2b] = 0;
  for (int k = 0; k < kNumBackward; ++k) {
    const CcStmt& s = kCcBackward[k];
    if (lambda_only && !s.flag) continue;
    if (kCcTensors[s.a].kind == K_BAR) { QTRY(ensure(s.a)); QTRY(settle(s.a)); }
    if (s.b >= 0 && kCcTensors[s.b].kind == K_BAR) { QTRY(ensure(s.b)); QTRY(settle(s.b)); }
    QTRY(run(s));
  }
  for (int k = 0; k < kNumTensors; ++k)
    if (kCcTensors[k].kind == K_BAR) QTRY(settle(k));        // never written in this sweep: really zero for the readers that follow
  return 0;
}

int CcLambda::kernel(const LambdaOptions& opt, int* n_iter, bool* converged) {
  const int64_t o = o_, v = v_, nov = o * v, N2 = o * o * v * v, na = nov + N2;
  const int t1b = id_of("t1_bar"), t2b = id_of("t2_bar");
  DeviceDIIS diis(std::max(opt.diis_space, 1), na);
  if (opt.diis_space > 1) QTRY(diis.init());
  *converged = false;
  int it = 0;
  for (it = 1; it <= opt.max_cycle; ++it) {
    QTRY(backward(true));
    // z_new = (dE/dt + (dn/dt)^T z) / D; the t2 part is projected on t2[i,j,a,b] = t2[j,i,b,a] (its antisymmetric
    // remainder multiplies a residual that vanishes identically and would only feed round-off into the ladder)
    QTRY(dcopy(nov, ptr_[t1b], zn_));
    QTRY(dcopy(N2, ptr_[t2b], zn_.p + nov));
    QTRY(axpby(N2, 0.0, zn_.p + nov, 0.5, zn_.p + nov));
    QTRY(perm4(zn_.p + nov, ptr_[t2b], o, o, v, v, 1, 0, 3, 2, 0.5, 1.0));
    QTRY(dev_div_denom(zn_, o, 1, v, 1, cc_.eo_, nullptr, cc_.ev_, nullptr));
    QTRY(dev_div_denom(zn_.p + nov, o, o, v, v, cc_.eo_, cc_.eo_, cc_.ev_, cc_.ev_));
    QTRY(dcopy(na, zn_, diff_)); QTRY(axpby(na, -1.0, z_, 1.0, diff_));
    QTRY(dev_dot(na, diff_, diff_, scal_));
    QTRY(dcopy(na, zn_, z_));
    if (it > 1 && opt.diis_space > 1) QTRY(diis.extrapolate(z_, diff_));
    double nn = 0.0;
    QTRY(dev_d2h(&nn, scal_, sizeof(double)));
    const double dz = std::sqrt(nn);
    if (opt.verbose > 0) std::fprintf(stderr, "[qemb lambda] cycle %3d  |dz| = %.3e\n", it, dz);
    if (!std::isfinite(dz)) { set_error("CCSD Lambda iteration diverged"); return QEMB_ERR_NUMERIC; }
    if (dz < opt.conv_tol) { *converged = true; break; }
  }
  *n_iter = it > opt.max_cycle ? opt.max_cycle : it;
  return 0;
}

int CcLambda::densities(double* dm1_mo, const double* T34, int nf, double* I_host) {
  const int64_t o = o_, v = v_, n = o + v;
  QTRY(backward(false));
  // ---- 1-RDM: (f_bar + f_bar^T)/2 + 2 on the occupied diagonal
  std::vector<double> foo((size_t)(o * o)), fov((size_t)(o * v)), fvv((size_t)(v * v));
  QTRY(dev_d2h(foo.data(), ptr_[id_of("dfoo_bar")], sizeof(double) * o * o));
  QTRY(dev_d2h(fov.data(), ptr_[id_of("fov_bar")], sizeof(double) * o * v));
  QTRY(dev_d2h(fvv.data(), ptr_[id_of("dfvv_bar")], sizeof(double) * v * v));
  if (dm1_mo) {
    std::fill(dm1_mo, dm1_mo + n * n, 0.0);
    for (int64_t i = 0; i < o; ++i) for (int64_t j = 0; j < o; ++j) dm1_mo[i * n + j] = 0.5 * (foo[i * o + j] + foo[j * o + i]);
    for (int64_t a = 0; a < v; ++a) for (int64_t b = 0; b < v; ++b) dm1_mo[(o + a) * n + o + b] = 0.5 * (fvv[a * v + b] + fvv[b * v + a]);
    for (int64_t i = 0; i < o; ++i) for (int64_t a = 0; a < v; ++a) dm1_mo[i * n + o + a] = dm1_mo[(o + a) * n + i] = 0.5 * fov[i * v + a];
    for (int64_t i = 0; i < o; ++i) dm1_mo[i * n + i] += 2.0;
  }
  if (!T34 || nf <= 0 || !I_host) return 0;
  // ---- dL/dV scattered into the full MO tensor, its four "one index in front" images, one GEMM with (P q'|r' s')
  const int64_t n2 = n * n, n4 = n2 * n2;
  DBuf Vb, V4, Id;
  QTRY(Vb.alloc(n4)); QTRY(dev_fill(Vb, n4, 0.0));
  struct Blk { const char* name; int off[4]; int64_t so[4]; };   // offsets (in units of o) and output strides per block index
  const int64_t s0 = n2 * n, s1 = n2, s2 = n, s3 = 1;
  const Blk blks[] = {
      {"oooo_bar", {0, 0, 0, 0}, {s0, s1, s2, s3}}, {"ovoo_bar", {0, 1, 0, 0}, {s0, s1, s2, s3}}, {"ovov_bar", {0, 1, 0, 1}, {s0, s1, s2, s3}},
      {"oovv_bar", {0, 0, 1, 1}, {s0, s1, s2, s3}}, {"ovvo_bar", {0, 1, 1, 0}, {s0, s1, s2, s3}}, {"ovvv_bar", {0, 1, 1, 1}, {s0, s1, s2, s3}},
      {"vvvv_l_bar", {1, 1, 1, 1}, {s0, s2, s1, s3}}};          // vvvv_l_bar[a,b,c,d] = dL/d(ac|bd)
  for (const Blk& b : blks) {
    const int id = id_of(b.name);
    if (id < 0 || !ptr_[id]) continue;
    Copy4Desc c{};
    const char* sig = kCcTensors[id].sig;
    int64_t stride = 1;
    for (int k = 3; k >= 0; --k) { c.dim[k] = sig[k] == 'o' ? o : v; c.si[k] = stride; stride *= c.dim[k]; }
    int64_t offset = 0;
    for (int k = 0; k < 4; ++k) { c.so[k] = b.so[k]; offset += (int64_t)b.off[k] * o * b.so[k]; }
    c.in = ptr_[id]; c.out = Vb.p + offset; c.alpha = 1.0; c.beta = 1.0;
    QTRY(dev_copy4(c));
  }
  QTRY(V4.alloc(n4));
  QTRY(dcopy(n4, Vb, V4));
  QTRY(perm4(V4, Vb, n, n, n, n, 1, 0, 2, 3, 1.0, 1.0));
  QTRY(perm4(V4, Vb, n, n, n, n, 2, 3, 0, 1, 1.0, 1.0));
  QTRY(perm4(V4, Vb, n, n, n, n, 3, 2, 0, 1, 1.0, 1.0));
  Vb.release();
  QTRY(Id.alloc(n * nf));
  QTRY(gemm(n, nf, n2 * n, 1.0, V4, n2 * n, true, T34, nf, false, 0.0, Id, nf));
  return dev_d2h(I_host, Id, sizeof(double) * n * nf);
}

}  // namespace qemb
