"""CPU: pins for the oracle's CCSD Lambda equations / response densities (solve_ccsd(relax=True),
molbe/solver.py:925-939).  PySCF is not installable and the reference has no golden for relax_density=True, so the
restatement is pinned by the definition of the quantities:
  * 1-RDM (correlation part) == d E_corr / d f   by finite differences of the re-converged CCSD energy,
  * 2-RDM (normal ordered)   == d E_corr / d V   along random 8-fold symmetric directions,
  * two electrons (CCSD exact): full 1-RDM == FCI 1-RDM,
  * multipliers = 0 reproduce the reference's unrelaxed expressions (shared/external/ccsd_rdm.py).
"""
import numpy as np
import pytest

from helpers import synthetic_fragment
from qemb_oracle import ccsd, ccsd_lambda, eri, rdm, scf


def _setup(n, o, seed, scale=None):
    h, e1 = synthetic_fragment(n, o, seed, scale=scale)
    mf = scf.rhf(h, e1, o)
    assert mf["converged"]
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    return h, e1, mf, eris


def _ecc(eris, **kw):
    conv, e, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-14, conv_tol_normt=1e-12, max_cycle=300, **kw)
    assert conv
    return e, t1, t2


def _eris_from_mo(eri_mo, fock, mo_energy, o):
    n = fock.shape[0]
    return ccsd.Eris(eri_mo, np.eye(n), o, mo_energy=mo_energy, fock=fock)


@pytest.mark.parametrize("n,o", [(5, 2), (6, 3)])
def test_densities_are_energy_derivatives(n, o):
    h, e1, mf, eris0 = _setup(n, o, 40 + n, scale=0.12)
    C = mf["mo_coeff"]
    eri_mo = eri.ao2mo_full(e1, C, compact=False)
    f0 = np.diag(mf["mo_energy"])
    eris = _eris_from_mo(eri_mo, f0, mf["mo_energy"], o)
    e0, t1, t2 = _ecc(eris)
    z1, z2, nit, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-12)
    assert lag.residual_norm() < 1e-10
    dm1, Vbar = ccsd_lambda.response_densities(lag, z1, z2)
    assert np.abs(dm1 - dm1.T).max() < 1e-14
    # (i) one-particle: dE_corr/df_pq (symmetric perturbation) = 2 dm1_corr[p,q] (p != q), dm1_corr[p,p]
    dm1c = dm1.copy(); dm1c[np.diag_indices(o)] -= 2.0
    eps = 1e-4
    rng = np.random.default_rng(1)
    for (p, q) in [(0, 1), (0, o), (1, n - 1), (o, n - 1), (o - 1, o - 1), (n - 1, n - 1), (o, o + 1 if o + 1 < n else o)]:
        W = np.zeros((n, n)); W[p, q] += 1.0
        if p != q:
            W[q, p] += 1.0
        ep = _ecc(_eris_from_mo(eri_mo, f0 + eps * W, mf["mo_energy"], o), t1=t1, t2=t2)[0]
        em = _ecc(_eris_from_mo(eri_mo, f0 - eps * W, mf["mo_energy"], o), t1=t1, t2=t2)[0]
        fd = (ep - em) / (2 * eps)
        assert abs(fd - np.sum(W * dm1c)) < 2e-7, (p, q, fd, np.sum(W * dm1c))
    # (ii) two-particle: directional derivative along random 8-fold symmetric W
    for k in range(3):
        B = rng.standard_normal((3, n, n)); B = B + B.transpose(0, 2, 1)
        W = np.einsum("Ppq,Prs->pqrs", B, B)
        W /= np.abs(W).max()
        ep = _ecc(_eris_from_mo(eri_mo + eps * W, f0, mf["mo_energy"], o), t1=t1, t2=t2)[0]
        em = _ecc(_eris_from_mo(eri_mo - eps * W, f0, mf["mo_energy"], o), t1=t1, t2=t2)[0]
        fd = (ep - em) / (2 * eps)
        an = np.sum(Vbar * W)
        assert abs(fd - an) < 5e-7 * max(1.0, abs(an)), (fd, an)
    # energy is recovered from the densities: E_corr = f.dm1_corr + sum Vbar o V  (L is linear in f and V)
    assert abs(np.sum(f0 * dm1c) + np.sum(Vbar * eri_mo) - e0) < 1e-10
    g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
    assert abs(0.5 * np.sum(g2 * eri_mo) - np.sum(Vbar * eri_mo)) < 1e-11


def test_zero_multipliers_give_the_unrelaxed_densities():
    h, e1, mf, eris = _setup(6, 2, 9, scale=0.12)
    e0, t1, t2 = _ecc(eris)
    lag = ccsd_lambda.Lagrangian(t1, t2, eris)
    dm1, Vbar = ccsd_lambda.response_densities(lag, np.zeros_like(t1), np.zeros_like(t2))
    assert np.abs(dm1 - rdm.make_rdm1_ccsd_t1(t1)).max() < 1e-14
    ref = 0.5 * rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
    assert np.abs(ccsd_lambda.symmetrise8(Vbar) - ccsd_lambda.symmetrise8(ref)).max() < 1e-14


def test_two_electron_response_density_is_the_fci_density():
    n, o = 5, 1
    h, e1, mf, eris = _setup(n, o, 7, scale=0.12)
    C = mf["mo_coeff"]
    e_cc, t1, t2 = _ecc(eris)
    z1, z2, _, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-13)
    dm1, _ = ccsd_lambda.response_densities(lag, z1, z2)
    h_mo = C.T @ h @ C
    eri_mo = eri.ao2mo_full(e1, C, compact=False)
    H = np.zeros((n * n, n * n))
    for p in range(n):
        for q in range(n):
            for r in range(n):
                for s in range(n):
                    H[p * n + q, r * n + s] = h_mo[p, r] * (q == s) + h_mo[q, s] * (p == r) + eri_mo[p, r, q, s]
    w, U = np.linalg.eigh(H)
    # lowest singlet (symmetric spatial function)
    for k in range(len(w)):
        c = U[:, k].reshape(n, n)
        if np.abs(c - c.T).max() < 1e-8:
            break
    assert abs(w[k] - (mf["e_tot"] + e_cc)) < 1e-10
    dm_fci = 2.0 * c @ c.T
    assert np.abs(dm1 - dm_fci).max() < 1e-8
