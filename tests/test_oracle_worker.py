"""The CPU-baseline worker (oracle/qemb_oracle/worker.py: the reference's run_solver, molbe/be_parallel.py:40-307, restated on the oracle's
pieces in BLAS form) against the oracle's reference path (dense einsum RCCSD + get_frag_energy)."""
import numpy as np
import pytest

from helpers import synthetic_fragment
from qemb_oracle import be, ccsd, eri, rdm, scf, worker


@pytest.mark.parametrize("n,o,nf", [(10, 3, 4), (14, 5, 5)])
def test_worker_matches_the_oracle_pipeline(n, o, nf):
    h, e1 = synthetic_fragment(n, o, 700 + n)
    rng = np.random.default_rng(n)
    sym = lambda a: a + a.T
    h1, veff0 = sym(rng.standard_normal((n, n))), sym(rng.standard_normal((n, n)))
    cen = (1.0, list(range(nf // 2)))
    s4 = eri.pack_s4(e1)
    out = worker.run_solver(h, None, s4, o, nf, cen, h1, veff0, eeval=True)
    mf = scf.rhf(h, e1, o)
    t1, t2, ecc, nit = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"])
    assert out["converged"] and abs(out["e_corr"] - ecc) < 1e-10 and abs(out["n_iter"] - nit) <= 1
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    C = mf["mo_coeff"]
    assert np.abs(out["rdm1_emb"] - C @ r1 @ C.T * 0.5).max() < 1e-8
    e_ref = be.get_frag_energy(C, o, nf, cen, np.zeros((n, n)), h1, r1, rdm.make_rdm2_urlx(t1, t2, with_dm1=False), s4, veff0, None, True)
    assert np.abs(np.array(out["e_f"]) - np.array(e_ref)).max() < 1e-8
