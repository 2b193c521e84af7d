"""Oracle for the Gamma-point periodic direct DF transform (test infrastructure).

Restates kbe/eri_onthefly.py:19-45 (`_j2c_cholesky_or_eig`) and :147-241 (the body of `integral_direct_DF` downstream of the integral
calls) in NumPy, in the reference's own order of operations: every plane-wave block is rotated into every fragment's space and
contracted with (L|G) there (:176-199), real-space blocks are rotated and added to their auxiliary rows (:201-217), the complex
fitted tensor is fitted with the Cholesky factor or the eigenvalue fallback (:219-227), `bb.T @ bb` WITHOUT conjugation (:229), the
1e-6 test of the imaginary part (:230-236) and restore('4') (:237).  The integral source is the same duck-typed object the device
path takes (quemb_amd/kbe_eri_onthefly.py).  Pinned by tests/golden/kbe_df.npz: outputs of the reference's own function run on the
same synthetic integrals (tests/golden/make_golden_kbe_df.py)."""
import numpy as np
import scipy.linalg as sla

from . import eri as oeri


def j2c_cholesky_or_eig(j2c):
    """:19-45"""
    try:
        return sla.cholesky(j2c, lower=True), True
    except sla.LinAlgError:
        d, V = sla.eigh(j2c)
        keep = d > 1e-14
        return V[:, keep] / np.sqrt(d[keep]) @ V[:, keep].conj().T, False


def integral_direct_DF(source, TAs, pw_step=7, aux_step=3):
    """-> (list of 4-fold packed fragment ERIs, ischol).  TAs: list of (nao, n) matrices."""
    naux = source.naux
    pqL = [np.zeros((naux, TA.shape[1], TA.shape[1]), dtype=np.complex128) for TA in TAs]        # :152-155
    fit, ischol = j2c_cholesky_or_eig(np.asarray(source.j2c()))                                   # :157-161
    nG = source.n_planewaves
    for g0 in range(0, nG, pw_step):                                                              # :176-199
        g1 = min(nG, g0 + pw_step)
        ints = np.asarray(source.pw_block(g0, g1))
        ft = np.asarray(source.ft_aux_block(g0, g1)).conj().T
        for k, TA in enumerate(TAs):
            Gqi = ints @ TA
            Gij = Gqi.transpose(0, 2, 1) @ TA.conj()
            pqL[k] += (ft @ Gij.reshape(g1 - g0, -1)).reshape(naux, TA.shape[1], TA.shape[1])
    for p0 in range(0, naux, aux_step):                                                           # :201-217
        p1 = min(naux, p0 + aux_step)
        ints = np.asarray(source.rs_block(p0, p1))
        for k, TA in enumerate(TAs):
            Lqi = ints @ TA
            pqL[k][p0:p1] += Lqi.transpose(0, 2, 1) @ TA.conj()
    out = []
    for k, TA in enumerate(TAs):                                                                  # :219-241
        n = TA.shape[1]
        b = pqL[k].reshape(naux, -1)
        bb = sla.solve_triangular(fit, b, lower=True) if ischol else fit @ b
        e = bb.T @ bb
        if (np.abs(e.imag) > 1e-6).any():
            raise ValueError(f"Imaginary part of ERI is larger than 1e-6 for frag #{k}.")
        out.append(oeri.pack_s4(e.real.reshape(n, n, n, n)))
    return out, ischol
