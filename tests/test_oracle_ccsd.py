"""CPU: known-answer tests for the PySCF-resident rows (a3, a4, a7, a8) of the oracle.

PySCF is not installable here, so the restatement of its RCCSD / RHF / ao2mo is pinned by independent physics:
  * RCCSD == an independently coded SPIN-ORBITAL CCSD (Stanton, Gauss, Watts, Bartlett JCP 94, 4334 (1991));
  * CCSD == exact FCI for two electrons;
  * DF transform with a complete auxiliary space == dense 4-index transform.
"""
import numpy as np
import pytest

from helpers import synthetic_fragment
from qemb_oracle import ccsd, eri, scf


# ---------------------------------------------------------------- independent spin-orbital CCSD
def spin_orbital_ccsd(h_mo, eri_mo, nocc, tol=1e-13, max_iter=400):
    n = h_mo.shape[0]
    ns = 2 * n
    # spin orbitals: index 2p = p alpha, 2p+1 = p beta; occupied first
    order = [2 * p + s for p in range(nocc) for s in (0, 1)] + [2 * p + s for p in range(nocc, n) for s in (0, 1)]
    sp = np.array([q // 2 for q in order]); ss = np.array([q % 2 for q in order])
    hs = h_mo[np.ix_(sp, sp)] * (ss[:, None] == ss[None, :])
    # <pq|rs> = (pr|qs) delta(sp,sr) delta(sq,ss)
    g = eri_mo[np.ix_(sp, sp, sp, sp)]                       # (pq|rs) chem over spatial parts
    g = g * (ss[:, None, None, None] == ss[None, :, None, None]) * (ss[None, None, :, None] == ss[None, None, None, :])
    phys = g.transpose(0, 2, 1, 3)                            # <pr|qs>... -> <pq|rs> = (pr|qs)
    asym = phys - phys.transpose(0, 1, 3, 2)
    no = 2 * nocc
    o, v = slice(0, no), slice(no, ns)
    f = hs + np.einsum("piqi->pq", asym[:, o, :, o])
    eo, ev = np.diag(f)[o], np.diag(f)[v]
    Dia = eo[:, None] - ev[None, :]
    Dijab = eo[:, None, None, None] + eo[None, :, None, None] - ev[None, None, :, None] - ev[None, None, None, :]
    t1 = np.zeros((no, ns - no)); t2 = asym[o, o, v, v] / Dijab
    E = 0.0
    for it in range(max_iter):
        tau_t = t2 + 0.5 * (np.einsum("ia,jb->ijab", t1, t1) - np.einsum("ib,ja->ijab", t1, t1))
        tau = t2 + np.einsum("ia,jb->ijab", t1, t1) - np.einsum("ib,ja->ijab", t1, t1)
        fvv, foo, fov = f[v, v], f[o, o], f[o, v]
        Fae = fvv - np.diag(np.diag(fvv)) - 0.5 * np.einsum("me,ma->ae", fov, t1) + np.einsum("mf,mafe->ae", t1, asym[o, v, v, v]) \
            - 0.5 * np.einsum("mnaf,mnef->ae", tau_t, asym[o, o, v, v])
        Fmi = foo - np.diag(np.diag(foo)) + 0.5 * np.einsum("ie,me->mi", t1, fov) + np.einsum("ne,mnie->mi", t1, asym[o, o, o, v]) \
            + 0.5 * np.einsum("inef,mnef->mi", tau_t, asym[o, o, v, v])
        Fme = fov + np.einsum("nf,mnef->me", t1, asym[o, o, v, v])
        Wmnij = asym[o, o, o, o] + np.einsum("je,mnie->mnij", t1, asym[o, o, o, v]) - np.einsum("ie,mnje->mnij", t1, asym[o, o, o, v]) \
            + 0.25 * np.einsum("ijef,mnef->mnij", tau, asym[o, o, v, v])
        Wabef = asym[v, v, v, v] - np.einsum("mb,amef->abef", t1, asym[v, o, v, v]) + np.einsum("ma,bmef->abef", t1, asym[v, o, v, v]) \
            + 0.25 * np.einsum("mnab,mnef->abef", tau, asym[o, o, v, v])
        Wmbej = asym[o, v, v, o] + np.einsum("jf,mbef->mbej", t1, asym[o, v, v, v]) - np.einsum("nb,mnej->mbej", t1, asym[o, o, v, o]) \
            - np.einsum("jnfb,mnef->mbej", 0.5 * t2 + np.einsum("jf,nb->jnfb", t1, t1), asym[o, o, v, v])
        r1 = fov + np.einsum("ie,ae->ia", t1, Fae) - np.einsum("ma,mi->ia", t1, Fmi) + np.einsum("imae,me->ia", t2, Fme) \
            - np.einsum("nf,naif->ia", t1, asym[o, v, o, v]) - 0.5 * np.einsum("imef,maef->ia", t2, asym[o, v, v, v]) \
            - 0.5 * np.einsum("mnae,nmei->ia", t2, asym[o, o, v, o])
        P = lambda x, perm: x - x.transpose(perm)
        r2 = asym[o, o, v, v].copy()
        tmp = np.einsum("ijae,be->ijab", t2, Fae - 0.5 * np.einsum("mb,me->be", t1, Fme)); r2 += P(tmp, (0, 1, 3, 2))
        tmp = np.einsum("imab,mj->ijab", t2, Fmi + 0.5 * np.einsum("je,me->mj", t1, Fme)); r2 -= P(tmp, (1, 0, 2, 3))
        r2 += 0.5 * np.einsum("mnab,mnij->ijab", tau, Wmnij) + 0.5 * np.einsum("ijef,abef->ijab", tau, Wabef)
        tmp = np.einsum("imae,mbej->ijab", t2, Wmbej) - np.einsum("ie,ma,mbej->ijab", t1, t1, asym[o, v, v, o])
        r2 += tmp - tmp.transpose(1, 0, 2, 3) - tmp.transpose(0, 1, 3, 2) + tmp.transpose(1, 0, 3, 2)
        tmp = np.einsum("ie,abej->ijab", t1, asym[v, v, v, o]); r2 += P(tmp, (1, 0, 2, 3))
        tmp = np.einsum("ma,mbij->ijab", t1, asym[o, v, o, o]); r2 -= P(tmp, (0, 1, 3, 2))
        t1n, t2n = r1 / Dia, r2 / Dijab
        En = np.einsum("ia,ia", fov, t1n) + 0.25 * np.einsum("ijab,ijab", asym[o, o, v, v], t2n) \
            + 0.5 * np.einsum("ijab,ia,jb", asym[o, o, v, v], t1n, t1n)
        d = max(np.abs(t1n - t1).max(), np.abs(t2n - t2).max())
        t1, t2 = t1n, t2n
        if abs(En - E) < tol and d < 1e-11:
            return En
        E = En
    raise RuntimeError("spin-orbital CCSD did not converge")


def _mo_problem(n, o, seed):
    h, e1 = synthetic_fragment(n, o, seed)
    mf = scf.rhf(h, e1, o)
    assert mf["converged"]
    C = mf["mo_coeff"]
    eri_mo = eri.ao2mo_full(e1, C, compact=False)
    h_mo = C.T @ h @ C
    return h, e1, mf, h_mo, eri_mo


@pytest.mark.parametrize("n,o", [(4, 1), (5, 2), (6, 3)])
def test_rccsd_equals_spin_orbital_ccsd(n, o):
    h, e1, mf, h_mo, eri_mo = _mo_problem(n, o, 100 + n)
    t1, t2, e_r, nit = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"], conv_tol=1e-13, conv_tol_normt=1e-11)
    e_so = spin_orbital_ccsd(h_mo, eri_mo, o)
    assert abs(e_r - e_so) < 1e-11, (e_r, e_so)
    assert np.abs(t2 - t2.transpose(1, 0, 3, 2)).max() < 1e-12


def test_ccsd_equals_fci_for_two_electrons():
    n, o = 5, 1
    h, e1, mf, h_mo, eri_mo = _mo_problem(n, o, 7)
    t1, t2, e_cc, _ = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"], conv_tol=1e-13, conv_tol_normt=1e-11)
    # singlet 2-electron FCI: symmetric spatial functions c_pq, H c = E c
    H = np.zeros((n * n, n * n))
    for p in range(n):
        for q in range(n):
            for r in range(n):
                for s in range(n):
                    H[p * n + q, r * n + s] = h_mo[p, r] * (q == s) + h_mo[q, s] * (p == r) + eri_mo[p, r, q, s]
    S = np.zeros((n * n, n * (n + 1) // 2)); k = 0
    for p in range(n):
        for q in range(p + 1):
            S[p * n + q, k] += 1.0; S[q * n + p, k] += 1.0
            S[:, k] /= np.linalg.norm(S[:, k]); k += 1
    e_fci = np.linalg.eigvalsh(S.T @ H @ S)[0]
    assert abs((mf["e_tot"] + e_cc) - e_fci) < 1e-10


def test_df_with_complete_aux_equals_dense():
    rng = np.random.default_rng(3)
    N, n = 6, 4
    npr = N * (N + 1) // 2
    # a complete "auxiliary basis": one function per AO pair, metric = the (exact) pair ERI matrix
    Bm = rng.standard_normal((npr + 3, N, N)); Bm = Bm + Bm.transpose(0, 2, 1)
    e1 = np.einsum("Ppq,Prs->pqrs", Bm, Bm)
    il = np.tril_indices(N)
    j2c = eri.pack_s4(e1)                                  # (P|Q), P,Q = AO pairs (SPD: rank npr)
    pqL = np.zeros((N, N, npr))
    pqL[il[0], il[1], :] = j2c; pqL[il[1], il[0], :] = j2c   # (mu nu|P)
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n]
    dense = eri.ao2mo_full(e1, TA)
    df = eri.integral_direct_DF(pqL, j2c, TA)
    assert np.abs(df - dense).max() < 1e-8 * np.abs(dense).max()
    packed = pqL[il[0], il[1], :].T                          # (naux, npair(N))
    df2 = eri.df_transform_packed(packed, np.linalg.cholesky(j2c), TA)
    assert np.abs(df2 - dense).max() < 1e-8 * np.abs(dense).max()


def test_rhf_is_a_fixed_point():
    h, e1 = synthetic_fragment(8, 3, 11)
    mf = scf.rhf(h, e1, 3)
    assert mf["converged"]
    C, F = mf["mo_coeff"], mf["fock"]
    assert np.abs(C.T @ F @ C - np.diag(mf["mo_energy"])).max() < 1e-9
    assert np.abs(C.T @ C - np.eye(8)).max() < 1e-12


def test_lean_update_matches_reference_form():
    """ccsd_lean.update_amps (bench cpu_baseline) == ccsd.update_amps on a DF-factorised fragment."""
    from qemb_oracle import ccsd_lean
    n, o = 9, 3
    rng = np.random.default_rng(21)
    B = 0.1 * rng.standard_normal((20, n, n)); B = 0.5 * (B + B.transpose(0, 2, 1))
    e1 = np.einsum("Ppq,Prs->pqrs", B, B)
    A = rng.standard_normal((n, n)); h = np.diag(2.0 * np.arange(n)) + 0.15 * (A + A.T)
    mf = scf.rhf(h, e1, o)
    C = mf["mo_coeff"]
    eris = ccsd.Eris(e1, C, o, mo_energy=mf["mo_energy"])
    lean = ccsd_lean.LeanEris(np.einsum("Ppq,pi,qj->Pij", B, C, C), o, mf["mo_energy"])
    t1, t2 = ccsd.init_amps(eris)
    t1 = t1 + 0.01 * rng.standard_normal(t1.shape)           # exercise the t1 terms
    a1, a2 = ccsd.update_amps(t1, t2, eris)
    b1, b2 = ccsd_lean.update_amps(t1, t2, lean)
    assert np.abs(a1 - b1).max() < 1e-12 and np.abs(a2 - b2).max() < 1e-12
