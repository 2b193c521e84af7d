/* gto_ints.c -- minimal Gaussian integral generator (Cartesian shells: s, p, d orbital functions; up to g auxiliary functions) for libqemb_gto.so.
 *
 * Role: the INTEGRAL SOURCE upstream of the hot path.  The reference obtains
 * hcore, S, (mu nu|ka la), (mu nu|P) and (P|Q) from PySCF/libcint (molbe/mbe.py:361-373,
 * molbe/eri_onthefly.py:64-108); neither is installed in this image, so real molecules (H8, octane in
 * STO-3G) would otherwise be out of reach and the reference's end-to-end golden energies unusable.
 * McMurchie-Davidson scheme: Hermite expansion coefficients E_t^{ij} and Hermite Coulomb integrals R_{tuv}
 * from the Boys function.  Host C + OpenMP: this is once-per-system CPU work in the reference as well.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXLPAIR 4       /* largest angular momentum of a function PAIR: d + d, or one g auxiliary function */
#define MAXPRIM 8
#define LTOT (2 * MAXLPAIR) /* max total angular momentum in an ERI: (dd|dd), (dd|g), (g|g) */

typedef struct {
  double ctr[3];
  int lmn[3];
  int nprim;
  double ex[MAXPRIM], co[MAXPRIM]; /* co includes primitive and contraction normalisation */
} bf_t;

/* Boys function F_0..F_nmax(t) */
static void boys(int nmax, double t, double* F) {
  if (t < 35.0) {
    /* series for F_nmax, then downward recursion */
    double term = 1.0 / (2 * nmax + 1), sum = term;
    for (int i = 1; i < 200; ++i) {
      term *= 2.0 * t / (2 * nmax + 2 * i + 1);
      sum += term;
      if (term < 1e-17 * sum) break;
    }
    const double et = exp(-t);
    F[nmax] = et * sum;
    for (int m = nmax; m > 0; --m) F[m - 1] = (2.0 * t * F[m] + et) / (2 * m - 1);
  } else {
    const double et = exp(-t);
    F[0] = 0.5 * sqrt(M_PI / t) * erf(sqrt(t));
    for (int m = 0; m < nmax; ++m) F[m + 1] = ((2 * m + 1) * F[m] - et) / (2.0 * t);
  }
}

/* Hermite expansion coefficients E[t], t = 0..i+j, for one Cartesian direction */
static double Ecoef(int i, int j, int t, double Qx, double a, double b) {
  const double p = a + b, q = a * b / p;
  if (t < 0 || t > i + j) return 0.0;
  if (i == 0 && j == 0 && t == 0) return exp(-q * Qx * Qx);
  if (j == 0)
    return (1.0 / (2 * p)) * Ecoef(i - 1, j, t - 1, Qx, a, b) - (q * Qx / a) * Ecoef(i - 1, j, t, Qx, a, b) +
           (t + 1) * Ecoef(i - 1, j, t + 1, Qx, a, b);
  return (1.0 / (2 * p)) * Ecoef(i, j - 1, t - 1, Qx, a, b) + (q * Qx / b) * Ecoef(i, j - 1, t, Qx, a, b) +
         (t + 1) * Ecoef(i, j - 1, t + 1, Qx, a, b);
}

/* R^0_{tuv} table for t+u+v <= L: R[t][u][v] */
#define RD (LTOT + 1)
static void rtable(int L, double p, const double PC[3], double R[RD][RD][RD]) {
  double F[LTOT + 1];
  static const int dummy = 0; (void)dummy;
  double Rn[LTOT + 1][RD][RD][RD];
  const double r2 = PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2];
  boys(L, p * r2, F);
  for (int n = 0; n <= L; ++n) Rn[n][0][0][0] = pow(-2.0 * p, n) * F[n];
  /* build by increasing total order, using the n+1 level */
  for (int N = 1; N <= L; ++N)
    for (int n = 0; n <= L - N; ++n)
      for (int t = 0; t <= N; ++t)
        for (int u = 0; u <= N - t; ++u) {
          const int v = N - t - u;
          double val;
          if (t > 0) {
            val = PC[0] * Rn[n + 1][t - 1][u][v];
            if (t > 1) val += (t - 1) * Rn[n + 1][t - 2][u][v];
          } else if (u > 0) {
            val = PC[1] * Rn[n + 1][t][u - 1][v];
            if (u > 1) val += (u - 1) * Rn[n + 1][t][u - 2][v];
          } else {
            val = PC[2] * Rn[n + 1][t][u][v - 1];
            if (v > 1) val += (v - 1) * Rn[n + 1][t][u][v - 2];
          }
          Rn[n][t][u][v] = val;
        }
  for (int t = 0; t <= L; ++t)
    for (int u = 0; u <= L - t; ++u)
      for (int v = 0; v <= L - t - u; ++v) R[t][u][v] = Rn[0][t][u][v];
}

static double prim_overlap(const double A[3], const int la[3], double a, const double B[3], const int lb[3], double b) {
  const double p = a + b;
  double s = pow(M_PI / p, 1.5);
  for (int d = 0; d < 3; ++d) s *= Ecoef(la[d], lb[d], 0, A[d] - B[d], a, b);
  return s;
}

static double prim_kinetic(const double A[3], const int la[3], double a, const double B[3], const int lb[3], double b) {
  const int L = lb[0] + lb[1] + lb[2];
  double t = b * (2 * L + 3) * prim_overlap(A, la, a, B, lb, b);
  for (int d = 0; d < 3; ++d) {
    int l2[3] = {lb[0], lb[1], lb[2]};
    l2[d] += 2;
    t += -2.0 * b * b * prim_overlap(A, la, a, B, l2, b);
    if (lb[d] >= 2) {
      l2[d] -= 4;
      t += -0.5 * lb[d] * (lb[d] - 1) * prim_overlap(A, la, a, B, l2, b);
    }
  }
  return t;
}

static double prim_nuclear(const double A[3], const int la[3], double a, const double B[3], const int lb[3], double b,
                           const double C[3]) {
  const double p = a + b;
  double P[3], PC[3];
  for (int d = 0; d < 3; ++d) { P[d] = (a * A[d] + b * B[d]) / p; PC[d] = P[d] - C[d]; }
  const int L = la[0] + la[1] + la[2] + lb[0] + lb[1] + lb[2];
  double R[RD][RD][RD];
  rtable(L, p, PC, R);
  double val = 0.0;
  for (int t = 0; t <= la[0] + lb[0]; ++t) {
    const double ex = Ecoef(la[0], lb[0], t, A[0] - B[0], a, b);
    for (int u = 0; u <= la[1] + lb[1]; ++u) {
      const double ey = Ecoef(la[1], lb[1], u, A[1] - B[1], a, b);
      for (int v = 0; v <= la[2] + lb[2]; ++v) val += ex * ey * Ecoef(la[2], lb[2], v, A[2] - B[2], a, b) * R[t][u][v];
    }
  }
  return 2.0 * M_PI / p * val;
}

/* one-electron matrices: S, T, V (V = sum_C -Z_C <a|1/r_C|b>) */
void gto_one_electron(int nbf, const bf_t* bf, int natm, const double* atm_xyz, const double* atm_Z, double* S, double* T,
                      double* V) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < nbf; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0, t = 0, v = 0;
      for (int pa = 0; pa < bf[i].nprim; ++pa)
        for (int pb = 0; pb < bf[j].nprim; ++pb) {
          const double c = bf[i].co[pa] * bf[j].co[pb];
          s += c * prim_overlap(bf[i].ctr, bf[i].lmn, bf[i].ex[pa], bf[j].ctr, bf[j].lmn, bf[j].ex[pb]);
          t += c * prim_kinetic(bf[i].ctr, bf[i].lmn, bf[i].ex[pa], bf[j].ctr, bf[j].lmn, bf[j].ex[pb]);
          for (int C = 0; C < natm; ++C)
            v -= c * atm_Z[C] * prim_nuclear(bf[i].ctr, bf[i].lmn, bf[i].ex[pa], bf[j].ctr, bf[j].lmn, bf[j].ex[pb], atm_xyz + 3 * C);
        }
      S[i * nbf + j] = S[j * nbf + i] = s;
      T[i * nbf + j] = T[j * nbf + i] = t;
      V[i * nbf + j] = V[j * nbf + i] = v;
    }
}

/* primitive-pair data: exponent sum, centre, Hermite coefficients per direction (t <= MAXLPAIR) times the contraction coefs */
typedef struct { double p, P[3], Ex[MAXLPAIR + 1], Ey[MAXLPAIR + 1], Ez[MAXLPAIR + 1], c; int tx, ty, tz; } ppair_t;

static int build_pairs(const bf_t* a, const bf_t* b, ppair_t* out) {
  int n = 0;
  for (int pa = 0; pa < a->nprim; ++pa)
    for (int pb = 0; pb < b->nprim; ++pb) {
      ppair_t* q = &out[n++];
      const double ea = a->ex[pa], eb = b->ex[pb];
      q->p = ea + eb;
      for (int d = 0; d < 3; ++d) q->P[d] = (ea * a->ctr[d] + eb * b->ctr[d]) / q->p;
      q->tx = a->lmn[0] + b->lmn[0]; q->ty = a->lmn[1] + b->lmn[1]; q->tz = a->lmn[2] + b->lmn[2];
      for (int t = 0; t <= MAXLPAIR; ++t) {
        q->Ex[t] = Ecoef(a->lmn[0], b->lmn[0], t, a->ctr[0] - b->ctr[0], ea, eb);
        q->Ey[t] = Ecoef(a->lmn[1], b->lmn[1], t, a->ctr[1] - b->ctr[1], ea, eb);
        q->Ez[t] = Ecoef(a->lmn[2], b->lmn[2], t, a->ctr[2] - b->ctr[2], ea, eb);
      }
      q->c = a->co[pa] * b->co[pb];
    }
  return n;
}

static double eri_from_pairs(const ppair_t* ab, int nab, const ppair_t* cd, int ncd) {
  double tot = 0.0;
  for (int x = 0; x < nab; ++x)
    for (int y = 0; y < ncd; ++y) {
      const ppair_t* A = &ab[x]; const ppair_t* B = &cd[y];
      const double p = A->p, q = B->p, alpha = p * q / (p + q);
      double PQ[3] = {A->P[0] - B->P[0], A->P[1] - B->P[1], A->P[2] - B->P[2]};
      const int L = A->tx + A->ty + A->tz + B->tx + B->ty + B->tz;
      double R[RD][RD][RD];
      rtable(L, alpha, PQ, R);
      double val = 0.0;
      for (int t = 0; t <= A->tx; ++t) for (int u = 0; u <= A->ty; ++u) for (int v = 0; v <= A->tz; ++v) {
        const double eab = A->Ex[t] * A->Ey[u] * A->Ez[v];
        if (eab == 0.0) continue;
        for (int tt = 0; tt <= B->tx; ++tt) for (int uu = 0; uu <= B->ty; ++uu) for (int vv = 0; vv <= B->tz; ++vv) {
          const double ecd = B->Ex[tt] * B->Ey[uu] * B->Ez[vv];
          if (ecd == 0.0) continue;
          const double sgn = ((tt + uu + vv) & 1) ? -1.0 : 1.0;
          val += eab * ecd * sgn * R[t + tt][u + uu][v + vv];
        }
      }
      tot += A->c * B->c * val * 2.0 * pow(M_PI, 2.5) / (p * q * sqrt(p + q));
    }
  return tot;
}

/* all (ab|cd) with 8-fold symmetry, written as the full nbf^4 tensor */
void gto_eri_s1(int nbf, const bf_t* bf, double* eri) {
  const long n = nbf, npair = n * (n + 1) / 2;
  ppair_t* pairs = (ppair_t*)malloc(sizeof(ppair_t) * npair * MAXPRIM * MAXPRIM);
  int* npp = (int*)malloc(sizeof(int) * npair);
#pragma omp parallel for schedule(dynamic)
  for (long ij = 0; ij < npair; ++ij) {
    long i = (long)((sqrt(8.0 * ij + 1.0) - 1.0) / 2.0);
    while (i * (i + 1) / 2 > ij) --i;
    while ((i + 1) * (i + 2) / 2 <= ij) ++i;
    const long j = ij - i * (i + 1) / 2;
    npp[ij] = build_pairs(&bf[i], &bf[j], pairs + ij * MAXPRIM * MAXPRIM);
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (long ij = 0; ij < npair; ++ij) {
    long i = (long)((sqrt(8.0 * ij + 1.0) - 1.0) / 2.0);
    while (i * (i + 1) / 2 > ij) --i;
    while ((i + 1) * (i + 2) / 2 <= ij) ++i;
    const long j = ij - i * (i + 1) / 2;
    for (long kl = 0; kl <= ij; ++kl) {
      long k = (long)((sqrt(8.0 * kl + 1.0) - 1.0) / 2.0);
      while (k * (k + 1) / 2 > kl) --k;
      while ((k + 1) * (k + 2) / 2 <= kl) ++k;
      const long l = kl - k * (k + 1) / 2;
      const double v = eri_from_pairs(pairs + ij * MAXPRIM * MAXPRIM, npp[ij], pairs + kl * MAXPRIM * MAXPRIM, npp[kl]);
#define SET(a, b, c, d) eri[(((a)*n + (b)) * n + (c)) * n + (d)] = v
      SET(i, j, k, l); SET(j, i, k, l); SET(i, j, l, k); SET(j, i, l, k);
      SET(k, l, i, j); SET(l, k, i, j); SET(k, l, j, i); SET(l, k, j, i);
#undef SET
    }
  }
  free(pairs); free(npp);
}

/* 3-centre (ab|P) and 2-centre (P|Q) Coulomb integrals with an auxiliary set of basis functions (for DF tests) */
void gto_eri_3c(int nbf, const bf_t* bf, int naux, const bf_t* aux, double* out /* nbf*nbf*naux */) {
  bf_t unit; memset(&unit, 0, sizeof(unit)); unit.nprim = 1; unit.ex[0] = 0.0; unit.co[0] = 1.0;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < nbf; ++i) {
    ppair_t ab[MAXPRIM * MAXPRIM], cd[MAXPRIM * MAXPRIM];
    for (int j = 0; j <= i; ++j) {
      const int nab = build_pairs(&bf[i], &bf[j], ab);
      for (int P = 0; P < naux; ++P) {
        bf_t u = unit; memcpy(u.ctr, aux[P].ctr, sizeof(u.ctr));
        const int ncd = build_pairs(&aux[P], &u, cd);
        const double v = eri_from_pairs(ab, nab, cd, ncd);
        out[((long)i * nbf + j) * naux + P] = out[((long)j * nbf + i) * naux + P] = v;
      }
    }
  }
}
void gto_eri_2c(int naux, const bf_t* aux, double* out /* naux*naux */) {
  bf_t unit; memset(&unit, 0, sizeof(unit)); unit.nprim = 1; unit.ex[0] = 0.0; unit.co[0] = 1.0;
#pragma omp parallel for schedule(dynamic)
  for (int P = 0; P < naux; ++P) {
    ppair_t ab[MAXPRIM * MAXPRIM], cd[MAXPRIM * MAXPRIM];
    bf_t u = unit; memcpy(u.ctr, aux[P].ctr, sizeof(u.ctr));
    const int nab = build_pairs(&aux[P], &u, ab);
    for (int Q = 0; Q <= P; ++Q) {
      bf_t w = unit; memcpy(w.ctr, aux[Q].ctr, sizeof(w.ctr));
      const int ncd = build_pairs(&aux[Q], &w, cd);
      out[(long)P * naux + Q] = out[(long)Q * naux + P] = eri_from_pairs(ab, nab, cd, ncd);
    }
  }
}

/* (ab|P) for a LIST of orbital pairs (the stored unique pairs of the semi-sparse tensor, molbe/eri_sparse_DF.py:410-494):
 * out[pair][P], npairs x naux row-major -- one auxiliary vector per pair, the layout of SemiSparseSym3DTensor.unique_dense_data */
void gto_eri_3c_pairs(int nbf, const bf_t* bf, int naux, const bf_t* aux, long npairs, const int* pi, const int* pj, double* out) {
  (void)nbf;
  bf_t unit; memset(&unit, 0, sizeof(unit)); unit.nprim = 1; unit.ex[0] = 0.0; unit.co[0] = 1.0;
  /* the auxiliary "pairs" (function x unit s function at the same centre) are the same for every orbital pair: build them once */
  ppair_t* auxp = (ppair_t*)malloc(sizeof(ppair_t) * (size_t)naux * MAXPRIM);
  int* nauxp = (int*)malloc(sizeof(int) * (size_t)naux);
  for (int P = 0; P < naux; ++P) {
    bf_t u = unit; memcpy(u.ctr, aux[P].ctr, sizeof(u.ctr));
    nauxp[P] = build_pairs(&aux[P], &u, auxp + (size_t)P * MAXPRIM);
  }
#pragma omp parallel for schedule(dynamic)
  for (long x = 0; x < npairs; ++x) {
    ppair_t ab[MAXPRIM * MAXPRIM];
    const int nab = build_pairs(&bf[pi[x]], &bf[pj[x]], ab);
    for (int P = 0; P < naux; ++P) out[x * naux + P] = eri_from_pairs(ab, nab, auxp + (size_t)P * MAXPRIM, nauxp[P]);
  }
  free(auxp); free(nauxp);
}

int gto_max_l_pair(void) { return MAXLPAIR; }
size_t gto_bf_size(void) { return sizeof(bf_t); }
