"""Oracle for row a8: the RCCSD executed by solve_ccsd (molbe/solver.py:829-946).  Test infrastructure.

The arithmetic lives in PySCF (pyscf>=2.0.0, pyproject.toml:20; not installed in this image).  This file
restates PySCF's published readable path -- cc/rccsd.py `update_amps` + cc/rintermediates.py -- exactly as
written in SURVEY.md Appendix A, with QuEmb's set-up: eris.fock = diag(mo_energy) (solver.py:901-902).
Pinned by tests/test_oracle_ccsd.py: == spin-orbital CCSD (Stanton-Gauss-Watts-Bartlett form) and == exact
FCI for two-electron systems.
"""
import numpy as np

from .eri import restore_s1


class Eris:
    """All MO-basis blocks in chemists' notation: ovov[i,a,j,b] = (ia|jb), ..."""

    def __init__(self, eri_emb, mo_coeff, nocc, mo_energy=None, fock=None):
        n = mo_coeff.shape[0]
        e1 = restore_s1(eri_emb, n)
        C = mo_coeff
        m = np.einsum("pqrs,pi->iqrs", e1, C, optimize=True)
        m = np.einsum("iqrs,qj->ijrs", m, C, optimize=True)
        m = np.einsum("ijrs,rk->ijks", m, C, optimize=True)
        m = np.einsum("ijks,sl->ijkl", m, C, optimize=True)
        o, v = slice(0, nocc), slice(nocc, n)
        self.nocc, self.nmo = nocc, n
        self.oooo = m[o, o, o, o].copy(); self.ovoo = m[o, v, o, o].copy(); self.ovov = m[o, v, o, v].copy()
        self.oovv = m[o, o, v, v].copy(); self.ovvo = m[o, v, v, o].copy(); self.ovvv = m[o, v, v, v].copy()
        self.vvvv = m[v, v, v, v].copy()
        self.mo_energy = None if mo_energy is None else np.asarray(mo_energy)
        self.fock = np.diag(self.mo_energy) if fock is None else fock


def energy(t1, t2, eris, es=None):
    nocc = eris.nocc
    fov = eris.fock[:nocc, nocc:]
    if es is not None:      # recording path (ccsd_lambda.py): same expression, differentiable
        tau = es("ia,jb->ijab", t1, t1) + t2
        return 2 * es("ia,ia->", fov, t1) + 2 * es("ijab,iajb->", tau, eris.ovov) - es("ijab,ibja->", tau, eris.ovov)
    e = 2 * np.einsum("ia,ia", fov, t1)
    tau = np.einsum("ia,jb->ijab", t1, t1) + t2
    e += 2 * np.einsum("ijab,iajb", tau, eris.ovov)
    e += -np.einsum("ijab,ibja", tau, eris.ovov)
    return float(e)


def init_amps(eris):
    nocc = eris.nocc
    eo, ev = eris.mo_energy[:nocc], eris.mo_energy[nocc:]
    eia = eo[:, None] - ev[None, :]
    eijab = eia[:, None, :, None] + eia[None, :, None, :]
    t1 = eris.fock[:nocc, nocc:] / eia
    t2 = eris.ovov.transpose(0, 2, 1, 3) / eijab
    return t1, t2


def _einsum(*a):
    return np.einsum(*a, optimize=True)


def amplitude_numerators(t1, t2, eris, es=_einsum):
    """SURVEY.md Appendix A, line by line (== pyscf cc/rccsd.py update_amps + rintermediates) up to the division by
    the orbital-energy denominators.  `es` is the contraction routine (ccsd_lambda.py passes a recording one)."""
    nocc, nvir = t1.shape
    fock = eris.fock
    eo, ev = eris.mo_energy[:nocc], eris.mo_energy[nocc:]
    fov = fock[:nocc, nocc:]; foo = fock[:nocc, :nocc]; fvv = fock[nocc:, nocc:]
    ovov, ovoo, ovvv, oovv, ovvo, oooo, vvvv = eris.ovov, eris.ovoo, eris.ovvv, eris.oovv, eris.ovvo, eris.oooo, eris.vvvv

    Foo = 2 * es("kcld,ilcd->ki", ovov, t2) - es("kdlc,ilcd->ki", ovov, t2) \
        + 2 * es("kcld,ic,ld->ki", ovov, t1, t1) - es("kdlc,ic,ld->ki", ovov, t1, t1) + foo
    Fvv = -2 * es("kcld,klad->ac", ovov, t2) + es("kdlc,klad->ac", ovov, t2) \
        - 2 * es("kcld,ka,ld->ac", ovov, t1, t1) + es("kdlc,ka,ld->ac", ovov, t1, t1) + fvv
    Fov = 2 * es("kcld,ld->kc", ovov, t1) - es("kdlc,ld->kc", ovov, t1) + fov
    Loo = Foo + es("kc,ic->ki", fov, t1) + 2 * es("lcki,lc->ki", ovoo, t1) - es("kcli,lc->ki", ovoo, t1)
    Lvv = Fvv - es("kc,ka->ac", fov, t1) + 2 * es("kdac,kd->ac", ovvv, t1) - es("kcad,kd->ac", ovvv, t1)
    Foo = Foo - np.diag(eo); Fvv = Fvv - np.diag(ev); Loo = Loo - np.diag(eo); Lvv = Lvv - np.diag(ev)

    t1new = -2 * es("kc,ka,ic->ia", fov, t1, t1) + es("ac,ic->ia", Fvv, t1) - es("ki,ka->ia", Foo, t1) \
        + 2 * es("kc,kica->ia", Fov, t2) - es("kc,ikca->ia", Fov, t2) + es("kc,ic,ka->ia", Fov, t1, t1) + fov \
        + 2 * es("kcai,kc->ia", ovvo, t1) - es("kiac,kc->ia", oovv, t1) \
        + 2 * es("kdac,ikcd->ia", ovvv, t2) - es("kcad,ikcd->ia", ovvv, t2) \
        + 2 * es("kdac,kd,ic->ia", ovvv, t1, t1) - es("kcad,kd,ic->ia", ovvv, t1, t1) \
        - 2 * es("lcki,klac->ia", ovoo, t2) + es("kcli,klac->ia", ovoo, t2) \
        - 2 * es("lcki,lc,ka->ia", ovoo, t1, t1) + es("kcli,lc,ka->ia", ovoo, t1, t1)

    tmp2 = es("kibc,ka->abic", oovv, -t1) + ovvv.transpose(1, 3, 0, 2)
    tmp = es("abic,jc->ijab", tmp2, t1)
    t2new = tmp + tmp.transpose(1, 0, 3, 2)
    tmp2 = es("kcai,jc->akij", ovvo, t1) + ovoo.transpose(1, 3, 0, 2)
    tmp = es("akij,kb->ijab", tmp2, t1)
    t2new -= tmp + tmp.transpose(1, 0, 3, 2)
    t2new += ovov.transpose(0, 2, 1, 3)

    Woooo = es("lcki,jc->klij", ovoo, t1) + es("kclj,ic->klij", ovoo, t1) + es("kcld,ijcd->klij", ovov, t2) \
        + es("kcld,ic,jd->klij", ovov, t1, t1) + oooo.transpose(0, 2, 1, 3)
    Wvvvv = es("kdac,kb->abcd", ovvv, -t1) - es("kcbd,ka->abcd", ovvv, t1) + vvvv.transpose(0, 2, 1, 3)
    Wvoov = es("kcad,id->akic", ovvv, t1) - es("kcli,la->akic", ovoo, t1) + ovvo.transpose(2, 0, 3, 1) \
        - 0.5 * es("ldkc,ilda->akic", ovov, t2) - 0.5 * es("lckd,ilad->akic", ovov, t2) \
        - es("ldkc,id,la->akic", ovov, t1, t1) + es("ldkc,ilad->akic", ovov, t2)
    Wvovo = es("kdac,id->akci", ovvv, t1) - es("lcki,la->akci", ovoo, t1) + oovv.transpose(2, 0, 3, 1) \
        - 0.5 * es("lckd,ilda->akci", ovov, t2) - es("lckd,id,la->akci", ovov, t1, t1)

    tau = t2 + es("ia,jb->ijab", t1, t1)
    t2new += es("klij,klab->ijab", Woooo, tau)
    t2new += es("abcd,ijcd->ijab", Wvvvv, tau)
    tmp = es("ac,ijcb->ijab", Lvv, t2); t2new += tmp + tmp.transpose(1, 0, 3, 2)
    tmp = es("ki,kjab->ijab", Loo, t2); t2new -= tmp + tmp.transpose(1, 0, 3, 2)
    tmp = 2 * es("akic,kjcb->ijab", Wvoov, t2) - es("akci,kjcb->ijab", Wvovo, t2)
    t2new += tmp + tmp.transpose(1, 0, 3, 2)
    tmp = es("akic,kjbc->ijab", Wvoov, t2); t2new -= tmp + tmp.transpose(1, 0, 3, 2)
    tmp = es("bkci,kjac->ijab", Wvovo, t2); t2new -= tmp + tmp.transpose(1, 0, 3, 2)
    return t1new, t2new


def update_amps(t1, t2, eris):
    nocc = t1.shape[0]
    eo, ev = eris.mo_energy[:nocc], eris.mo_energy[nocc:]
    t1new, t2new = amplitude_numerators(t1, t2, eris)
    eia = eo[:, None] - ev[None, :]
    eijab = eia[:, None, :, None] + eia[None, :, None, :]
    return t1new / eia, t2new / eijab


class DIIS:
    """pyscf lib.diis.DIIS semantics for CC amplitudes (space 6): the error vector of a trial vector is its
    difference from the previously RETURNED (extrapolated) vector."""

    def __init__(self, space=6):
        self.space = space
        self.xs, self.es = [], []
        self.xprev = None

    def update(self, x):
        if self.xprev is None and not self.xs:
            self.xprev = x.copy()
            return x
        self.xs.append(x.copy()); self.es.append(x - self.xprev)
        if len(self.xs) > self.space:
            self.xs.pop(0); self.es.pop(0)
        m = len(self.xs)
        B = np.zeros((m + 1, m + 1)); B[0, 1:] = B[1:, 0] = 1.0
        for i in range(m):
            for j in range(m):
                B[i + 1, j + 1] = self.es[i] @ self.es[j]
        rhs = np.zeros(m + 1); rhs[0] = 1.0
        try:
            c = np.linalg.solve(B, rhs)[1:]
        except np.linalg.LinAlgError:
            c = np.linalg.lstsq(B, rhs, rcond=None)[0][1:]
        xn = sum(ci * xi for ci, xi in zip(c, self.xs))
        self.xprev = xn.copy()
        return xn


def kernel(eris, conv_tol=1e-10, conv_tol_normt=1e-8, max_cycle=100, diis_space=6, t1=None, t2=None):
    """pyscf cc/ccsd.py `kernel` control flow (MP2 guess, DIIS from cycle 0, |dE| < tol and |dt| < tolnormt).
    PySCF defaults are conv_tol 1e-7 / 1e-5 / 50 cycles; the oracle converges tighter.  Returns
    (converged, e_corr, t1, t2, n_iter)."""
    if t1 is None or t2 is None:
        t1, t2 = init_amps(eris)
    nocc, nvir = t1.shape
    ecc = energy(t1, t2, eris)
    adiis = DIIS(diis_space)
    conv = False
    it = 0
    for it in range(1, max_cycle + 1):
        t1n, t2n = update_amps(t1, t2, eris)
        normt = np.sqrt(np.linalg.norm(t1n - t1) ** 2 + np.linalg.norm(t2n - t2) ** 2)
        vec = adiis.update(np.concatenate([t1n.ravel(), t2n.ravel()]))
        t1 = vec[: nocc * nvir].reshape(nocc, nvir); t2 = vec[nocc * nvir:].reshape(nocc, nocc, nvir, nvir)
        eold, ecc = ecc, energy(t1, t2, eris)
        if abs(ecc - eold) < conv_tol and normt < conv_tol_normt:
            conv = True
            break
    return conv, ecc, t1, t2, it


def solve_ccsd(h1, eri_emb, nocc, mo_coeff, mo_energy, **kw):
    """solve_ccsd (molbe/solver.py:829-946) given the fragment RHF result: returns (t1, t2, e_corr, n_iter)."""
    eris = Eris(eri_emb, mo_coeff, nocc, mo_energy=mo_energy)
    conv, e, t1, t2, it = kernel(eris, **kw)
    if not conv:
        raise RuntimeError("oracle CCSD did not converge")
    return t1, t2, e, it
