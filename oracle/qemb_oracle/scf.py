"""Oracle for rows a6 / a7 (fragment Fock construction and fragment RHF).  Test infrastructure."""
import numpy as np

from .eri import restore_s1


def get_jk(eri1, dm):
    """scf.hf.dot_eri_dm semantics (molbe/helper.py:64): J_pq = (pq|rs) D_rs, K_pr = (pq|rs) D_qs."""
    J = np.einsum("pqrs,rs->pq", eri1, dm, optimize=True)
    K = np.einsum("pqrs,qs->pr", eri1, dm, optimize=True)
    return J, K


def get_veff(eri, dm, S, TA, hf_veff):
    """Restates molbe/helper.py:28-69.  Returns (Veff, Veff0)."""
    n = TA.shape[1]
    ST = S @ TA
    P_ = ST.T @ dm @ ST
    e1 = restore_s1(eri, n)
    vj, vk = get_jk(e1, P_)
    Veff_ = vj - 0.5 * vk
    Veff0 = TA.T @ hf_veff @ TA
    return Veff0 - Veff_, Veff0


def rhf(h1, eri, nocc, dm0=None, max_cycle=50, conv_tol=1e-12, conv_tol_grad=1e-8, diis_space=8, jk=None):
    """Restates get_scfObj (molbe/helper.py:73-151): closed-shell RHF in an orthonormal basis (S = I) with
    hcore = h1, 8-fold ERIs, nelec = 2*nocc, optional dm0, <= 50 cycles, commutator DIIS.  PySCF's
    default conv_tol is 1e-9; the oracle converges tighter so that it is a fixed point, not a trajectory.
    `jk(dm) -> (J, K)` replaces the dense contraction for fragments whose n^4 tensor should not be formed (eri may then be None).
    Returns dict(mo_coeff, mo_energy, mo_occ, e_tot, converged, dm)."""
    n = h1.shape[0]
    if jk is None:
        e1 = restore_s1(eri, n)
        jk = lambda dm: get_jk(e1, dm)
    if dm0 is None:
        w, c = np.linalg.eigh(h1)
        dm = 2.0 * c[:, :nocc] @ c[:, :nocc].T
    else:
        dm = np.array(dm0, dtype=float)
    fs, es = [], []
    e_old = None
    conv = False
    for cyc in range(max_cycle):
        J, K = jk(dm)
        F = h1 + J - 0.5 * K
        e_tot = 0.5 * np.einsum("ij,ji->", h1 + F, dm)
        err = F @ dm - dm @ F
        gnorm = np.linalg.norm(err)
        if e_old is not None and abs(e_tot - e_old) < conv_tol and gnorm < conv_tol_grad:
            conv = True
            break
        e_old = e_tot
        fs.append(F.copy()); es.append(err.copy())
        if len(fs) > diis_space:
            fs.pop(0); es.pop(0)
        Fd = F
        if len(fs) > 1:
            m = len(fs)
            B = np.zeros((m + 1, m + 1)); B[-1, :] = B[:, -1] = 1.0; B[-1, -1] = 0.0
            for i in range(m):
                for j in range(m):
                    B[i, j] = np.vdot(es[i], es[j])
            rhs = np.zeros(m + 1); rhs[-1] = 1.0
            try:
                c = np.linalg.solve(B, rhs)[:m]
                Fd = sum(ci * fi for ci, fi in zip(c, fs))
            except np.linalg.LinAlgError:
                Fd = F
        w, c = np.linalg.eigh(Fd)
        dm = 2.0 * c[:, :nocc] @ c[:, :nocc].T
    # canonical orbitals of the converged Fock matrix
    J, K = jk(dm)
    F = h1 + J - 0.5 * K
    w, c = np.linalg.eigh(F)
    occ = np.zeros(n); occ[:nocc] = 2.0
    return dict(mo_coeff=c, mo_energy=w, mo_occ=occ, e_tot=0.5 * np.einsum("ij,ji->", h1 + F, dm), converged=conv,
                dm=dm, fock=F, cycles=cyc + 1)
