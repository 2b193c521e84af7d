"""Fixtures for the analytic MP2 / CCSD-model Jacobians (`jac_solver="MP2" | "CCSD"`): tests/golden/jac.npz.

The reference builds the initial quasi-Newton Jacobian from per-fragment density responses dP/d(lambda):
  "HF"   -> hfres_func   (shared/external/optqn.py:424-434):   CPHF (cphf_utils.py:55-81)
  "MP2"  -> mp2res_func  (optqn.py:437-447):  get_dPmp2_batch_r (cpmp2_utils.py:94-133), halved
  "CCSD" -> ccsdres_func (optqn.py:450-461):  get_dPccsdurlx_batch_u (jac_utils.py:162-178) -- the HF response plus the derivative of
            an APPROXIMATE t1 (MP2 doubles put through one cycle of the CCSD t1 equation, jac_utils.py:13-41); not the exact CCSD response.
This script CALLS those reference functions (run in the build container only; /root/reference is absent on the GPU box).  They reach
PySCF for two things only -- `ao2mo.incore.general(V, (C1, C2, C3, C4))`, the four-index transformation (pq|rs) C1_pi C2_qj C3_rk C4_sl,
and `scf.hf.dot_eri_dm(V, dm)`, J_pq = (pq|rs) dm_rs, K_pr = (pq|rs) dm_qs -- and PySCF is not installed, so the stub modules that stand
in for the missing third-party roots (tests/golden/make_golden.py) serve exactly these two definitions from einsum on the dense tensor.
Everything else that runs is the reference's own code.  Output: inputs (C, mo_energy, eri, n_occ, potentials) and the three response sets.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import make_golden as mg  # noqa: E402


def _dense(V, n):
    V = np.asarray(V)
    if V.ndim == 4:
        return V
    from qemb_oracle import eri as oeri
    return oeri.restore_s1(V, n)


def install_pyscf_pieces():
    import pyscf  # the stub package installed by make_golden._install()
    from pyscf import ao2mo, scf

    def general(V, Cs, compact=False):
        n = Cs[0].shape[0]
        e = _dense(V, n)
        out = np.einsum("pqrs,pi,qj,rk,sl->ijkl", e, Cs[0], Cs[1], Cs[2], Cs[3], optimize=True)
        return out.reshape(Cs[0].shape[1] * Cs[1].shape[1], Cs[2].shape[1] * Cs[3].shape[1])

    def dot_eri_dm(V, dm, hermi=0, with_j=True, with_k=True):
        e = _dense(V, dm.shape[0])
        return np.einsum("pqrs,rs->pq", e, dm), np.einsum("pqrs,qs->pr", e, dm)

    ao2mo.incore.general = general
    scf.hf.dot_eri_dm = dot_eri_dm
    return pyscf


def cases():
    from helpers import synthetic_fragment
    from qemb_oracle import scf as oscf
    out = []
    for (n, o, seed, edges) in ((7, 3, 11, [[0, 1]]), (10, 4, 12, [[0, 1], [4, 5]]), (9, 2, 13, [[2, 3, 4]])):
        h, e1 = synthetic_fragment(n, o, seed, scale=0.12)
        mf = oscf.rhf(h, e1, o, conv_tol=1e-13, conv_tol_grad=1e-9)
        assert mf["converged"]
        vp = []
        for e in edges:
            for j in range(len(e)):
                for k in range(j, len(e)):
                    m = np.zeros((n, n)); m[e[j], e[k]] = m[e[k], e[j]] = 1.0
                    vp.append(m)
        m = np.zeros((n, n))
        members = set(x for e in edges for x in e)
        for f in range(min(6, n)):
            if f not in members:
                m[f, f] = -1.0
        vp.append(m)
        out.append(dict(n=n, o=o, h=h, eri=e1, C=mf["mo_coeff"], moe=mf["mo_energy"], vpots=np.array(vp)))
    return out


def main():
    mg._install()
    install_pyscf_pieces()
    from quemb.shared.external import cphf_utils, cpmp2_utils, jac_utils
    data = {}
    for c, case in enumerate(cases()):
        C, moe, e1, o, vp = case["C"], case["moe"], case["eri"], case["o"], list(case["vpots"])
        us = cphf_utils.cphf_kernel_batch(C, moe, e1, o, vp)
        dP_hf = np.array([cphf_utils.get_rhf_dP_from_u(C, o, u) for u in us])
        dP_mp2 = np.array([0.5 * d for d in cpmp2_utils.get_dPmp2_batch_r(C, moe, e1, o, vp, aorep=True)])     # mp2res_func
        dP_cc = np.array(jac_utils.get_dPccsdurlx_batch_u(C, moe, e1, o, vp))                                  # ccsdres_func
        for k, v in dict(n=case["n"], o=o, h=case["h"], eri=e1, C=C, moe=moe, vpots=case["vpots"], us=np.array(us), dP_hf=dP_hf,
                         dP_mp2=dP_mp2, dP_ccsd=dP_cc).items():
            data[f"c{c}_{k}"] = v
        print(f"case {c}: n={case['n']} o={o} npot={len(vp)} |dP_hf|={np.abs(dP_hf).max():.4f} |dP_mp2|={np.abs(dP_mp2).max():.4f} |dP_ccsd|={np.abs(dP_cc).max():.4f}")
    np.savez(HERE / "jac.npz", **data)


if __name__ == "__main__":
    main()
