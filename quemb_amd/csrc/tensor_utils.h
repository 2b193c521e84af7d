// tensor_utils.h -- small host-side helpers shared by the drivers: RAII device buffers, tensor
// permutation / block extraction on top of dev_copy4, GEMM shorthands, and a device-resident DIIS.
// Plain C++ over dev_ops.h (no HIP here).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "dev_ops.h"

namespace qemb {

#define QTRY(expr)              \
  do {                          \
    int _rc = (expr);           \
    if (_rc != 0) return _rc;   \
  } while (0)

struct DBuf {
  double* p = nullptr;
  int64_t n = 0;
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~DBuf() { release(); }
  int alloc(int64_t nelem) {
    release();
    void* q = nullptr;
    int rc = dev_alloc(&q, (size_t)(nelem > 0 ? nelem : 1) * sizeof(double));
    if (rc) return rc;
    p = (double*)q; n = nelem;
    return 0;
  }
  void release() { if (p) { dev_free(p); p = nullptr; n = 0; } }
  operator double*() const { return p; }
};

// y = alpha*x + beta*y over n contiguous elements
inline int axpby(int64_t n, double alpha, const double* x, double beta, double* y) {
  const int64_t C = 1 << 20;
  int64_t done = 0;
  while (done < n) {
    const int64_t left = n - done;
    Copy4Desc c{};
    if (left >= C) {
      int64_t rows = left / C;
      if (rows > (1 << 10)) rows = 1 << 10;
      c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = rows; c.dim[3] = C;
    } else {
      c.dim[0] = 1; c.dim[1] = 1; c.dim[2] = 1; c.dim[3] = left;
    }
    c.in = x + done; c.out = y + done;
    c.si[0] = c.si[1] = 0; c.si[2] = c.dim[3]; c.si[3] = 1;
    c.so[0] = c.so[1] = 0; c.so[2] = c.dim[3]; c.so[3] = 1;
    c.alpha = alpha; c.beta = beta;
    QTRY(dev_copy4(c));
    done += c.dim[2] * c.dim[3];
  }
  return 0;
}
inline int dcopy(int64_t n, const double* x, double* y) { return axpby(n, 1.0, x, 0.0, y); }
// out = a*x + b*y in one pass (out may alias x or y)
inline int lincomb2(int64_t n, double a, const double* x, double b, const double* y, double* out) {
  const double c[2] = {a, b};
  const double* xs[2] = {x, y};
  return dev_lincomb(n, 2, c, xs, 0.0, out);
}

// dst (contiguous, dims d[perm[0..3]]) = alpha * transpose(src, perm) + beta * dst;  src contiguous dims d
inline int perm4(double* dst, const double* src, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int p0, int p1, int p2,
                 int p3, double alpha = 1.0, double beta = 0.0, const double* base = nullptr) {
  const int64_t d[4] = {d0, d1, d2, d3};
  const int perm[4] = {p0, p1, p2, p3};
  int64_t od[4], ostr[4];
  for (int k = 0; k < 4; ++k) od[k] = d[perm[k]];
  ostr[3] = 1; ostr[2] = od[3]; ostr[1] = od[3] * od[2]; ostr[0] = od[3] * od[2] * od[1];
  Copy4Desc c{};
  c.in = src; c.out = dst; c.alpha = alpha; c.beta = beta;
  c.si[3] = 1; c.si[2] = d[3]; c.si[1] = d[3] * d[2]; c.si[0] = d[3] * d[2] * d[1];
  for (int k = 0; k < 4; ++k) { c.dim[k] = d[k]; }
  for (int k = 0; k < 4; ++k) c.so[perm[k]] = ostr[k];
  c.base = base;      // dst = alpha * transpose(src) + beta * base  (base laid out like dst; nullptr: accumulate into dst)
  return dev_copy4(c);
}

// dst (contiguous s0 x s1 x s2 x s3) = src[o0:o0+s0, o1:o1+s1, o2:.., o3:..] of a contiguous n0 x n1 x n2 x n3 tensor
inline int extract4(double* dst, const double* src, int64_t n1, int64_t n2, int64_t n3, int64_t o0, int64_t o1, int64_t o2,
                    int64_t o3, int64_t s0, int64_t s1, int64_t s2, int64_t s3) {
  Copy4Desc c{};
  c.dim[0] = s0; c.dim[1] = s1; c.dim[2] = s2; c.dim[3] = s3;
  c.si[3] = 1; c.si[2] = n3; c.si[1] = n3 * n2; c.si[0] = n3 * n2 * n1;
  c.in = src + o0 * c.si[0] + o1 * c.si[1] + o2 * c.si[2] + o3;
  c.so[3] = 1; c.so[2] = s3; c.so[1] = s3 * s2; c.so[0] = s3 * s2 * s1;
  c.out = dst; c.alpha = 1.0; c.beta = 0.0;
  return dev_copy4(c);
}

// C(MxN, ldc) = alpha * A * B + beta * C with explicit storage flags (see dev_ops.h GemmDesc)
inline int gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, bool a_kc, const double* B,
                int64_t ldb, bool b_kc, double beta, double* C, int64_t ldc, int64_t batch = 1, int64_t sA = 0,
                int64_t sB = 0, int64_t sC = 0, int cfg = -1, int ksplit = 0) {
  GemmDesc g{};
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.A = A; g.lda = lda; g.a_kcontig = a_kc ? 1 : 0; g.strideA = sA;
  g.B = B; g.ldb = ldb; g.b_kcontig = b_kc ? 1 : 0; g.strideB = sB;
  g.C = C; g.ldc = ldc; g.strideC = sC; g.batch = batch; g.cfg = cfg; g.ksplit = ksplit;
  return dev_gemm(g);
}
// row-major conveniences: A is (M x K) or, transposed, stored (K x M); B is (K x N) or stored (N x K)
inline int gemm_nn(int64_t M, int64_t N, int64_t K, double al, const double* A, const double* B, double be, double* C) {
  return gemm(M, N, K, al, A, K, true, B, N, false, be, C, N);
}
inline int gemm_nt(int64_t M, int64_t N, int64_t K, double al, const double* A, const double* B, double be, double* C) {
  return gemm(M, N, K, al, A, K, true, B, K, true, be, C, N);
}
inline int gemm_tn(int64_t M, int64_t N, int64_t K, double al, const double* A, const double* B, double be, double* C) {
  return gemm(M, N, K, al, A, M, false, B, N, false, be, C, N);
}

// Quarter transform whose result is symmetric in its first two indices and of which only the lower rows are kept:
//   Out[x'][y'][rest] = sum_x C[x,x'] In[y'][rest][x]   for x' >= y' only   (x', y' < n; `inner` = length of rest; x < K; C is K x n row-major)
// Block by block over y': for the columns of y' in [s0, s1) the rows x' < s0 are not computed.  The boundaries s0 = n - T walk down the
// row-tile heights T that exist as tile configurations, so every block fills its tiles: 61 % of the MFMA work of the full product at
// n = 220 (the operand In is still streamed once).  Entries with x' < y' of Out are left untouched.  Outside 192 < n <= 224 (where the
// 224-row tile holds all rows) the full product on the dispatcher's tile choice.
inline int gemm_quarter_lower_rows(int n, int64_t inner, int64_t K, const double* C, const double* In, double* Out) {
  const int64_t ncol = (int64_t)n * inner;
  if (!(n > 192 && n <= 224)) return gemm(n, ncol, K, 1.0, C, n, false, In, K, true, 0.0, Out, ncol);
  static const struct { int rows, cfg; } tiles[] = {{224, 13}, {192, 15}, {128, 4}, {112, 33}, {64, 12}};
  int s0 = 0;
  for (int t = 0; t < 5 && s0 < n; ++t) {
    const int s1 = (t + 1 < 5) ? std::min(n, std::max(s0, n - tiles[t + 1].rows)) : n;
    if (s1 > s0) {
      const int rc = gemm(n - s0, (int64_t)(s1 - s0) * inner, K, 1.0, C + s0, n, false, In + (int64_t)s0 * inner * K, K, true, 0.0,
                          Out + (int64_t)s0 * ncol + (int64_t)s0 * inner, ncol, 1, 0, 0, 0, tiles[t].cfg);
      if (rc) return rc;
    }
    s0 = s1;
  }
  return 0;
}

// small dense solve (Gaussian elimination with partial pivoting) for the DIIS equations, host side
inline bool solve_dense(int n, std::vector<double>& A, std::vector<double>& b) {
  for (int k = 0; k < n; ++k) {
    int piv = k; double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[i * n + k]) > best) { best = std::fabs(A[i * n + k]); piv = i; }
    if (best < 1e-300) return false;
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]); std::swap(b[k], b[piv]); }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * b[j];
    b[k] = s / A[k * n + k];
  }
  return true;
}

// Device-resident DIIS over vectors of length n.  Error vectors are supplied by the caller.
// (CCSD: error = trial - previously returned vector, PySCF lib.diis.DIIS semantics; SCF: FD - DF.)
class DeviceDIIS {
 public:
  DeviceDIIS(int space, int64_t n) : space_(space), n_(n) {}
  int init() {
    xs_.resize(space_); es_.resize(space_);
    for (int i = 0; i < space_; ++i) { QTRY(xs_[i].alloc(n_)); QTRY(es_[i].alloc(n_)); }
    QTRY(scal_.alloc(space_ + 1));
    B_.assign((size_t)space_ * space_, 0.0);
    return 0;
  }
  int size() const { return count_ < space_ ? count_ : space_; }
  // push (x, e) and overwrite x with the extrapolated vector
  int extrapolate(double* x, const double* e) {
    QTRY(dcopy(n_, x, next_x()));
    QTRY(dcopy(n_, e, next_e()));
    return extrapolate_pushed(x, true);
  }
  // Zero-copy variant for the large vectors of the CCSD iteration: the producer writes the trial vector and its error vector
  // straight into the storage of the next slot, then extrapolate_pushed(x_out) forms x_out = sum_i c_i x_i.
  double* next_x() { return xs_[count_ % space_]; }
  double* next_e() { return es_[count_ % space_]; }
  // err_dot (optional): <e, e> of the vector just pushed -- the diagonal of the Gram row this call computes anyway
  int extrapolate_pushed(double* x, bool x_holds_trial = false, double* err_dot = nullptr) {
    QTRY(gram_issue());
    QTRY(dev_sync());
    return gram_finish(x, x_holds_trial, err_dot);
  }
  // The same in two halves, so that a caller driving several solvers can put ONE wait between them for all:
  // gram_issue: the new row of the Gram matrix (one pass over the pushed error vector per eight stored ones) on its way to pinned host memory;
  // gram_finish (after a dev_sync of this context): the small solve on the host and x = sum_i c_i x_i.
  int gram_issue() {
    const int slot = count_ % space_;
    ++count_;
    const int m = size();
    QTRY(ensure_row_host());
    flagged_ = false;
    // refresh row/column `slot` of the Gram matrix
    for (int j0 = 0; j0 < m; j0 += 8) {   // one pass over the new error vector per eight stored ones
      const int cnt = std::min(8, m - j0);
      const double* ps[8];
      for (int q = 0; q < cnt; ++q) ps[q] = es_[j0 + q].p;
      QTRY(dev_dot_many(n_, es_[slot], cnt, ps, scal_.p + j0));
    }
    QTRY(dev_d2h_async(row_host_, scal_.p, sizeof(double) * m));
    pending_slot_ = slot;
    return 0;
  }
  // The push for error vectors that are DIFFERENCES (e = trial - prev, the CCSD iteration), fused: one launch forms e in the next slot, stores
  // the trial vector there (unless the producer wrote it in place: trial == next_x()) and sends the new Gram row to pinned host memory.
  int push_diff_issue(const double* trial, const double* prev) {
    const int slot = count_ % space_;
    if (space_ > 8) {
      QTRY(lincomb2(n_, 1.0, trial, -1.0, prev, es_[slot]));
      if (trial != xs_[slot].p) QTRY(dcopy(n_, trial, xs_[slot]));
      return gram_issue();
    }
    ++count_;
    const int m = size();
    QTRY(ensure_row_host());
    const double* ps[8];
    for (int q = 0; q < m; ++q) ps[q] = es_[q].p;
    QTRY(dev_diis_push(n_, trial, prev, es_[slot], trial == xs_[slot].p ? nullptr : xs_[slot].p, m, ps, slot, scal_.p, row_host_, row_host_ + space_ + 1, ++seq_));
    pending_slot_ = slot; flagged_ = true;
    return 0;
  }
  // wait for the row of the last push: on the word the fused launch publishes after it (microseconds), else on the stream
  int wait_row() {
    if (flagged_) { flagged_ = false; return dev_wait_flag(row_host_ + space_ + 1, seq_); }
    return dev_sync();
  }
  // After a wait: the Gram matrix takes the new row, the small system is solved on the host.  c / xs (room for `space` entries) receive the
  // terms of x = sum_i c_i x_i; when there is nothing to extrapolate (one vector, singular or non-finite system) that is the trial vector itself.
  int coefficients(int* nterms, double* c, const double** xs, double* err_dot = nullptr) {
    const int slot = pending_slot_;
    *nterms = 1; c[0] = 1.0; xs[0] = xs_[slot].p;
    const int m = size();
    const double* row = row_host_;
    for (int j = 0; j < m; ++j) { B_[(size_t)slot * space_ + j] = row[j]; B_[(size_t)j * space_ + slot] = row[j]; }
    if (err_dot) *err_dot = row[slot];
    if (m < 2) return 0;
    std::vector<double> A((size_t)(m + 1) * (m + 1), 0.0), rhs(m + 1, 0.0);
    double scale = 0.0;
    for (int i = 0; i < m; ++i) scale = std::fmax(scale, B_[(size_t)i * space_ + i]);
    if (scale <= 0.0) return 0;
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < m; ++j) A[(size_t)(i + 1) * (m + 1) + (j + 1)] = B_[(size_t)i * space_ + j] / scale;
      A[i + 1] = 1.0; A[(size_t)(i + 1) * (m + 1)] = 1.0;
    }
    rhs[0] = 1.0;
    if (!solve_dense(m + 1, A, rhs)) return 0;   // singular: keep the un-extrapolated vector
    for (int i = 0; i < m; ++i) if (!std::isfinite(rhs[i + 1])) return 0;
    *nterms = m;
    for (int i = 0; i < m; ++i) { c[i] = rhs[i + 1]; xs[i] = xs_[i].p; }
    return 0;
  }
  int gram_finish(double* x, bool x_holds_trial = false, double* err_dot = nullptr) {
    std::vector<double> c((size_t)space_);
    std::vector<const double*> ps((size_t)space_);
    int m = 0;
    QTRY(coefficients(&m, c.data(), ps.data(), err_dot));
    if (m == 1) return x_holds_trial ? 0 : dcopy(n_, ps[0], x);        // nothing to extrapolate: the trial vector
    for (int i0 = 0; i0 < m; i0 += 8)     // x = sum_i c_i x_i in one pass per eight vectors
      QTRY(dev_lincomb(n_, std::min(8, m - i0), c.data() + i0, ps.data() + i0, i0 == 0 ? 0.0 : 1.0, x));
    return 0;
  }
  int space() const { return space_; }
  void reset() { count_ = 0; }

 private:
  int space_; int64_t n_; int count_ = 0;
  std::vector<DBuf> xs_, es_;
  DBuf scal_;
  std::vector<double> B_;
  double* row_host_ = nullptr;      // pinned: the Gram row on its way back (gram_issue -> gram_finish), space + 1 doubles, then the word the fused push publishes
  unsigned long long seq_ = 0;      // ... its expected value (the block is zeroed when it is taken: recycled pinned blocks carry old words)
  bool flagged_ = false;
  int pending_slot_ = 0;
  int ensure_row_host() {
    if (row_host_) return 0;
    void* q = nullptr;
    QTRY(dev_pinned_alloc(&q, sizeof(double) * (space_ + 2)));
    row_host_ = (double*)q;
    for (int i = 0; i < space_ + 2; ++i) row_host_[i] = 0.0;
    seq_ = 0;
    return 0;
  }
 public:
  ~DeviceDIIS() { if (row_host_) { if (flagged_) (void)dev_sync_device(); dev_pinned_free(row_host_); } }      // (flagged_: a push was issued and never waited for -- an error exit)
  DeviceDIIS(DeviceDIIS&& o) noexcept : space_(o.space_), n_(o.n_), count_(o.count_), xs_(std::move(o.xs_)), es_(std::move(o.es_)), scal_(std::move(o.scal_)),
                                        B_(std::move(o.B_)), row_host_(o.row_host_), seq_(o.seq_), flagged_(o.flagged_), pending_slot_(o.pending_slot_) { o.row_host_ = nullptr; }
  DeviceDIIS(const DeviceDIIS&) = delete;
  DeviceDIIS& operator=(const DeviceDIIS&) = delete;
  DeviceDIIS& operator=(DeviceDIIS&&) = delete;
};

}  // namespace qemb
