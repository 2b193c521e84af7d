"""CPU restatement of the reference's pool worker `run_solver` (molbe/be_parallel.py:40-307) for solver == "CCSD": ONE whole fragment
from the inputs the worker receives to the outputs it sends back.  Test / measurement infrastructure (bench.py's `cpu_baseline` leg and
tests/test_oracle_worker.py) -- never imported by the product.

    inputs  (be_parallel.py:40-60):   h = h1 + heff (fock + heff), dm0, nao, nocc, n_frag, weight_and_relAO_per_center, TA, h1_e, the
                                      fragment's dataset `f{I}` of eri_file.h5 (4-fold packed ERIs), veff, veff0, eeval
    steps   (:108-307):               get_scfObj (fragment RHF, helper.py:73-151) -> solve_ccsd (solver.py:829-946: mycc.ao2mo(),
                                      mycc.kernel() with DIIS) -> make_rdm1 (ccsd_rdm.py:10) -> back-rotation (be_parallel.py:266-274)
                                      -> get_frag_energy (helper.py:220-339, with the unrelaxed 2-RDM of ccsd_rdm.py:17-55)
    outputs (:301-307):               (e_f[3], mo_coeff, rdm1_emb, rdm1_mo)  (+ n_iter here)

The steps are the oracle's own pieces; the amplitude equations run in their BLAS form (ccsd_lean.update_amps == ccsd.update_amps,
tests/test_oracle_ccsd.py) on MO blocks cut out of the dense transformed tensor, because that is what makes n ~ 130 affordable on a CPU.
"""
import numpy as np

from . import be, ccsd, ccsd_lean, rdm, scf


def unpack_s4(eri_s4, n):
    """ao2mo.restore(1, eri, n) (helper.py:189): the dense (n,n,n,n) tensor from the 4-fold packed block, as two gathers"""
    i, j = np.indices((n, n))
    P = (np.maximum(i, j) * (np.maximum(i, j) + 1) // 2 + np.minimum(i, j)).ravel()
    return np.ascontiguousarray(np.asarray(eri_s4)[P][:, P]).reshape(n, n, n, n)


def rotate4(T, C):
    """sum_{pqrs} T[p,q,r,s] C[p,i] C[q,j] C[r,k] C[s,l] as four BLAS quarter transforms without strided copies of the n^4 tensor
    (the operation of mycc.ao2mo() and of the 2-RDM back-rotation einsum at helper.py:287-295)"""
    n = C.shape[0]
    X = C.T @ T.reshape(n, -1)                                  # [i, (q r s)]
    X = np.matmul(C.T, X.reshape(n, n, n * n))                  # [i, j, (r s)]
    X = np.matmul(C.T, X.reshape(n * n, n, n))                  # [(i j), k, s]
    return np.matmul(X, C).reshape(n, n, n, n)                  # [(i j), k, l]


def mo_blocks(e1, C, o):
    """mycc.ao2mo(): the full four-index transformation to the fragment's canonical orbitals, cut into the blocks update_amps reads."""
    n = C.shape[0]
    m = rotate4(e1, C)
    oc, vi = slice(0, o), slice(o, n)
    cp = np.ascontiguousarray
    Vl = cp(m[vi, vi, vi, vi].transpose(0, 2, 1, 3))         # V[a,b,c,d] = (ac|bd)
    return dict(oooo=cp(m[oc, oc, oc, oc]), ovoo=cp(m[oc, vi, oc, oc]), ovov=cp(m[oc, vi, oc, vi]), oovv=cp(m[oc, oc, vi, vi]),
                ovvo=cp(m[oc, vi, vi, oc]), ovvv=cp(m[oc, vi, vi, vi]), Vl=Vl)


def frag_energy(C, nocc, n_frag, weight_and_relAO_per_center, h1, rdm1, rdm2s, e1, veff0, veff=None, use_cumulant=True):
    """get_frag_energy (helper.py:220-339) with the same arithmetic as be.get_frag_energy, organised for speed: the 2-RDM is rotated to the
    embedding basis in full, as the reference does (:287-295), by BLAS quarter transforms; the loop over (i, j) with the symmetrised G_ij
    against row P(ij) of the packed ERIs (:303-325) is the full contraction sum_jkl G[i,j,k,l] (ij|kl) with the unpacked tensor."""
    rdm1s_rot = C @ rdm1 @ C.T * 0.5
    hf = C[:, :nocc] @ C[:, :nocc].T
    if use_cumulant:
        delta = 2 * (rdm1s_rot - hf)
        e1_ = np.einsum("ij,ij->i", h1[:n_frag], delta[:n_frag])
        ec = np.einsum("ij,ij->i", veff0[:n_frag], delta[:n_frag])
    else:
        e1_ = 2 * np.einsum("ij,ij->i", h1[:n_frag], rdm1s_rot[:n_frag])
        ec = np.einsum("ij,ij->i", veff[:n_frag], rdm1s_rot[:n_frag])
    r2 = rotate4(0.5 * rdm2s, C.T)
    n = C.shape[0]
    e2 = np.einsum("ix,ix->i", r2.reshape(n, -1)[:n_frag], e1.reshape(n, -1)[:n_frag])
    w, cen = weight_and_relAO_per_center
    return [sum(w * e1_[i] for i in cen), sum(w * e2[i] for i in cen), sum(w * ec[i] for i in cen)]


def run_solver(h, dm0, eri_s4, nocc, n_frag, weight_and_relAO_per_center, h1, veff0, veff=None, eeval=True, use_cumulant=True,
               conv_tol=1e-10, conv_tol_normt=1e-8, max_cycle=100, scf_conv_tol=1e-11):
    n = h.shape[0]
    e1 = unpack_s4(eri_s4, n)
    eK = np.ascontiguousarray(e1.transpose(0, 2, 1, 3)).reshape(n * n, n * n)        # [(p r), (q s)]: the exchange contraction as a matvec
    eJ = e1.reshape(n * n, n * n)
    jk = lambda dm: ((eJ @ dm.ravel()).reshape(n, n), (eK @ dm.ravel()).reshape(n, n))
    mf = scf.rhf(h, None, nocc, dm0=dm0, conv_tol=scf_conv_tol, conv_tol_grad=1e-7, jk=jk)
    del eK
    C, eps = mf["mo_coeff"], mf["mo_energy"]
    b = mo_blocks(e1, C, nocc)
    er = ccsd_lean.LeanEris.from_blocks(nocc, eps, **b)
    conv, ecc, t1, t2, nit = ccsd_lean.kernel(er, conv_tol=conv_tol, conv_tol_normt=conv_tol_normt, max_cycle=max_cycle)
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    rdm1_emb = C @ r1 @ C.T * 0.5
    e_f = None
    if eeval:
        r2 = rdm.make_rdm2_urlx(t1, t2, with_dm1=not use_cumulant)
        e_f = frag_energy(C, nocc, n_frag, weight_and_relAO_per_center, h1, r1, r2, e1, veff0, veff, use_cumulant)
    return dict(e_f=e_f, mo_coeff=C, rdm1_emb=rdm1_emb, rdm1_mo=r1, e_corr=ecc, n_iter=nit, converged=bool(conv and mf["converged"]),
                scf_cycles=mf["cycles"], e_scf=mf["e_tot"])
