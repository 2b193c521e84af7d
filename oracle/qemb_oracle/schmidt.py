"""Oracle for rows a1 / a1' / a2 (Schmidt decomposition).  Test infrastructure."""
import numpy as np


def schmidt_decomposition(mo_coeff, nocc, AO_in_frag, thr_bath=1.0e-10, rdm=None):
    """Restates molbe/pfrag.py:403-494 (default branch, norb=None, cinv=None).

    D = C_occ C_occ^T (:448-450); Denv = D[env, env] (:465); eigh (:468); bath = eigenvectors with
    thr < |lambda| < 1 - thr (:484-486); TA = [I_frag (+) Evec_bath] (:489-491).
    Returns (TA_lo_eo, n_f, n_b).
    """
    if rdm is None:
        C = np.asarray(mo_coeff)[:, :nocc]
        Dhf = C @ C.T
    else:
        Dhf = np.asarray(rdm)
    N = Dhf.shape[0]
    frag = list(AO_in_frag)
    fset = set(frag)
    env = np.array([i for i in range(N) if i not in fset], dtype=int)
    Denv = Dhf[np.ix_(env, env)]
    ev, evec = np.linalg.eigh(Denv)
    bidx = [i for i in range(len(ev)) if thr_bath < abs(ev[i]) < 1.0 - thr_bath]
    TA = np.zeros((N, len(frag) + len(bidx)))
    TA[frag, np.arange(len(frag))] = 1.0
    TA[env, len(frag):] = evec[:, bidx]
    return TA, len(frag), len(bidx)


def schmidt_decomp_svd(rdm, Frag_sites, thr_bath=1.0e-10):
    """Restates kbe/solver.py:9-46: SVD of D[env, frag]; bath = left vectors with sigma >= thr (:41)."""
    rdm = np.asarray(rdm)
    N = rdm.shape[0]
    frag = [i if i >= 0 else N + i for i in Frag_sites]
    fset = set(frag)
    env = np.array([i for i in range(N) if i not in fset], dtype=int)
    Denv = rdm[np.ix_(env, frag)]
    U, sigma, _ = np.linalg.svd(Denv, full_matrices=False)
    nb = int((sigma >= thr_bath).sum())
    TA = np.zeros((N, len(frag) + nb), dtype=rdm.dtype)
    TA[frag, np.arange(len(frag))] = 1.0
    TA[env, len(frag):] = U[:, :nb]
    return TA


def get_nsocc(TA, S, C, nocc, ncore=0):
    """Restates molbe/pfrag.py:208-239: C_ = TA^T S C_occ; nsocc = round(tr(C_ C_^T)); initial fragment MOs
    = left singular vectors of C_ (:233).  Returns (P_, nsocc, mo_coeffs)."""
    C_ = TA.T @ S @ C[:, ncore:ncore + nocc]
    P_ = C_ @ C_.T
    nsocc = int(round(np.trace(P_)))
    mo = np.linalg.svd(C_)[0]
    return P_, nsocc, mo
