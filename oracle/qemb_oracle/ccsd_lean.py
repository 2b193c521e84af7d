"""Memory-lean CPU form of one RCCSD amplitude update for LARGE fragments (bench.py's cpu_baseline leg).

Same equations as `ccsd.update_amps`; differences are purely organisational so that a
(20 occupied, 200 virtual) fragment fits in host memory: integrals are assembled from the density-fitting factor
of the synthetic fragment family, the (vv|vv) block is stored once in the ladder layout V[a,b,c,d] = (ac|bd), the
t1-dressing of Wvvvv is folded on the tau side instead of building a second v^4 tensor, and every O(N^6) term is a
BLAS GEMM through tensordot/matmul.  Checked against `ccsd.update_amps` in tests/test_oracle_ccsd.py.
Test / measurement infrastructure only.
"""
import numpy as np


class LeanEris:
    @classmethod
    def from_blocks(cls, nocc, mo_energy, oooo, ovoo, ovov, oovv, ovvo, ovvv, Vl):
        """Wrap already assembled MO blocks (e.g. exported from the device, bench.py)."""
        self = cls.__new__(cls)
        self.nocc, self.nmo = nocc, len(mo_energy)
        self.mo_energy = np.asarray(mo_energy)
        self.oooo, self.ovoo, self.ovov, self.oovv, self.ovvo, self.ovvv, self.Vl = oooo, ovoo, ovov, oovv, ovvo, ovvv, Vl
        return self

    def __init__(self, B_mo, nocc, mo_energy):
        """B_mo: (naux, n, n) DF factor in the MO basis: (pq|rs) = sum_P B[P,p,q] B[P,r,s]."""
        o = nocc
        n = B_mo.shape[1]
        v = n - o
        self.nocc, self.nmo = o, n
        self.mo_energy = np.asarray(mo_energy)
        Boo, Bov, Bvv = B_mo[:, :o, :o], B_mo[:, :o, o:], B_mo[:, o:, o:]
        P = B_mo.shape[0]
        f = lambda x: np.ascontiguousarray(x).reshape(P, -1)
        self.oooo = (f(Boo).T @ f(Boo)).reshape(o, o, o, o)
        self.ovoo = (f(Bov).T @ f(Boo)).reshape(o, v, o, o)
        self.ovov = (f(Bov).T @ f(Bov)).reshape(o, v, o, v)
        self.oovv = (f(Boo).T @ f(Bvv)).reshape(o, o, v, v)
        self.ovvo = (f(Bov).T @ f(np.transpose(Bov, (0, 2, 1)))).reshape(o, v, v, o)
        self.ovvv = (f(Bov).T @ f(Bvv)).reshape(o, v, v, v)
        # V[a,b,c,d] = (ac|bd), built slab by slab (no second v^4 temporary)
        self.Vl = np.empty((v, v, v, v))
        Bvv_f = f(Bvv)                                     # [P, (b d)]
        for a in range(v):
            slab = Bvv[:, a, :].T @ Bvv_f                  # [c, (b d)]
            self.Vl[a] = slab.reshape(v, v, v).transpose(1, 0, 2)


def update_amps(t1, t2, er):
    o, v = t1.shape
    eo, ev = er.mo_energy[:o], er.mo_energy[o:]
    ovov, ovoo, ovvv, oovv, ovvo, oooo = er.ovov, er.ovoo, er.ovvv, er.oovv, er.ovvo, er.oooo
    nov = o * v
    tau = t2 + np.einsum("ia,jb->ijab", t1, t1)
    Lov = 2 * ovov - ovov.transpose(0, 3, 2, 1)                       # [k,c,l,d]
    Loovv = Lov.transpose(0, 2, 1, 3)                                  # [k,l,c,d]
    Foo = Loovv.reshape(o, -1) @ tau.reshape(o, -1).T                  # Foo'[k,i]
    Fvv = -np.tensordot(tau, Loovv, axes=([0, 1, 3], [0, 1, 3]))       # Fvv'[a,c] = -tau[klad] L[klcd]
    Fov = (Lov.reshape(nov, nov) @ t1.ravel()).reshape(o, v)
    Lovoo = 2 * ovoo - ovoo.transpose(2, 1, 0, 3)                      # [l,c,k,i]
    Z = np.tensordot(t1, Lovoo, axes=([0, 1], [0, 1]))                 # [k,i]
    Y = 2 * np.tensordot(t1, ovvv, axes=([0, 1], [0, 1])) - np.einsum("kcad,kd->ac", ovvv, t1, optimize=True)
    Loo, Lvv = Foo + Z, Fvv + Y
    T = t2.transpose(0, 2, 1, 3)                                       # [k,c,j,b] = t2[k,j,c,b]
    Tp = t2.transpose(0, 3, 1, 2)                                      # [k,c,j,b] = t2[k,j,b,c]
    # ---- T1
    t1n = t1 @ Lvv.T - Loo.T @ t1 + (t1 @ Fov.T) @ t1
    t1n += ((2 * T - Tp).reshape(nov, nov) @ Fov.ravel()).reshape(o, v)
    Lph1 = 2 * ovvo.transpose(3, 2, 0, 1) - oovv.transpose(1, 2, 0, 3)  # [(ia),(kc)]
    t1n += (Lph1.reshape(nov, nov) @ t1.ravel()).reshape(o, v)
    Th = 2 * t2.transpose(0, 1, 3, 2) - t2                              # [i,k,d,c] = 2 t2[ikcd] - t2[ikdc]
    t1n += Th.reshape(o, -1) @ ovvv.reshape(-1, v)
    t1n -= Lovoo.reshape(-1, o).T @ np.ascontiguousarray(T).reshape(-1, v)
    # ---- T2, direct part
    t2n = ovov.transpose(0, 2, 1, 3).copy()
    OV = ovov.transpose(0, 2, 1, 3).reshape(o * o, v * v)
    Wo = oooo.transpose(0, 2, 1, 3).reshape(o * o, o * o) + OV @ tau.reshape(o * o, -1).T
    O1 = np.einsum("jc,lcki->ljki", t1, ovoo, optimize=True)
    Wo = Wo + (O1.transpose(2, 0, 3, 1) + O1.transpose(0, 2, 1, 3)).reshape(o * o, o * o)
    t2n += (Wo.T @ tau.reshape(o * o, -1)).reshape(o, o, v, v)
    t2n += (tau.reshape(o * o, -1) @ er.Vl.reshape(v * v, v * v).T).reshape(o, o, v, v)      # pp-ladder
    # ---- P(X) part
    U = np.einsum("ac,ijcb->ijab", Lvv, t2, optimize=True) - np.einsum("ki,kjab->ijab", Loo, t2, optimize=True)
    X = tau.reshape(o * o, -1) @ ovvv.transpose(0, 2, 3, 1).reshape(nov, -1).T                # X[(ij),(k,a)] = tau[ijcd] ovvv[kdac]
    U -= np.einsum("ijka,kb->ijab", X.reshape(o, o, o, v), t1, optimize=True)
    G = np.einsum("jc,iabc->jiab", t1, ovvv, optimize=True); U += G.transpose(1, 0, 2, 3)
    G = np.einsum("kibc,jc->kibj", oovv, t1, optimize=True); U -= np.einsum("ka,kibj->ijab", t1, G, optimize=True)
    U -= np.einsum("iajk,kb->ijab", ovoo, t1, optimize=True)
    G = np.einsum("jc,kcai->kjai", t1, ovvo, optimize=True); U -= np.einsum("kjai,kb->ijab", G, t1, optimize=True)
    W1 = ovvo.transpose(3, 2, 0, 1) + np.einsum("kcad,id->iakc", ovvv, t1, optimize=True) - np.einsum("kcli,la->iakc", ovoo, t1, optimize=True)
    S = T - 0.5 * Tp - np.einsum("id,la->iald", t1, t1)
    ovov_t = ovov.transpose(0, 3, 2, 1)
    W1 = W1.reshape(nov, nov) + S.reshape(nov, nov) @ ovov.reshape(nov, nov) - 0.5 * np.ascontiguousarray(T).reshape(nov, nov) @ np.ascontiguousarray(ovov_t).reshape(nov, nov)
    W2 = oovv.transpose(1, 2, 0, 3) + np.einsum("id,kdac->iakc", t1, ovvv, optimize=True) - np.einsum("la,lcki->iakc", t1, ovoo, optimize=True)
    S = 0.5 * Tp + np.einsum("id,la->iald", t1, t1)
    W2 = W2.reshape(nov, nov) - S.reshape(nov, nov) @ np.ascontiguousarray(ovov_t).reshape(nov, nov)
    Tm, Tpm = np.ascontiguousarray(T).reshape(nov, nov), np.ascontiguousarray(Tp).reshape(nov, nov)
    R = (2 * W1 - W2) @ Tm - W1 @ Tpm
    U += R.reshape(o, v, o, v).transpose(0, 2, 1, 3)
    R = W2 @ Tpm
    U -= R.reshape(o, v, o, v).transpose(0, 2, 3, 1)
    t2n += U + U.transpose(1, 0, 3, 2)
    eia = eo[:, None] - ev[None, :]
    return t1n / eia, t2n / (eia[:, None, :, None] + eia[None, :, None, :])


def init_amps(er):
    """MP2 amplitudes (pyscf cc/ccsd.py init_amps with a diagonal Fock matrix): t1 = 0, t2 = ovov / D."""
    o = er.nocc
    eo, ev = er.mo_energy[:o], er.mo_energy[o:]
    eia = eo[:, None] - ev[None, :]
    return np.zeros((o, len(ev))), er.ovov.transpose(0, 2, 1, 3) / (eia[:, None, :, None] + eia[None, :, None, :])


def energy(t1, t2, er):
    tau = t2 + np.einsum("ia,jb->ijab", t1, t1)
    ov = er.ovov.transpose(0, 2, 1, 3)
    return float(2.0 * np.sum(ov * tau) - np.sum(ov.transpose(0, 1, 3, 2) * tau))


def kernel(er, conv_tol=1e-10, conv_tol_normt=1e-8, max_cycle=100, diis_space=6):
    """`ccsd.kernel` (pyscf cc/ccsd.py control flow: MP2 guess, DIIS on [t1; t2] from the first cycle, |dE| and |dt| tests) on the
    lean update.  Returns (converged, e_corr, t1, t2, n_iter)."""
    from .ccsd import DIIS
    t1, t2 = init_amps(er)
    o, v = t1.shape
    ecc = energy(t1, t2, er)
    adiis = DIIS(diis_space)
    conv, it = False, 0
    for it in range(1, max_cycle + 1):
        t1n, t2n = update_amps(t1, t2, er)
        normt = np.sqrt(np.linalg.norm(t1n - t1) ** 2 + np.linalg.norm(t2n - t2) ** 2)
        vec = adiis.update(np.concatenate([t1n.ravel(), t2n.ravel()]))
        t1 = vec[: o * v].reshape(o, v); t2 = vec[o * v:].reshape(o, o, v, v)
        eold, ecc = ecc, energy(t1, t2, er)
        if abs(ecc - eold) < conv_tol and normt < conv_tol_normt:
            conv = True
            break
    return conv, ecc, t1, t2, it
