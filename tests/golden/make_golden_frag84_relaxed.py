"""Fixture for the -m gpu RELAXED-density test at the benchmark's n_occ = 20 (tests/test_gpu_fragment.py::test_relaxed_fragment_at_bench_tiles):
the fragment of make_golden_frag84.py (n = 84, n_occ = 20, n_virt = 64: the ladder / ring / dressing kernel instantiations of BASELINE
configs[2]) with relax_density = 1 -- solve_ccsd(relax=True), molbe/solver.py:925-939.  The oracle: RCCSD to 1e-11, the Lambda equations
by the reverse-mode restatement (oracle/qemb_oracle/ccsd_lambda.py, pinned by energy derivatives and FCI in tests/test_oracle_lambda.py),
response 1-RDM, relaxed 2-RDM, get_frag_energy.  Minutes of NumPy, hence stored.

    python tests/golden/make_golden_frag84_relaxed.py        (writes tests/golden/frag84_relaxed.npz)
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests" / "golden"))
import make_golden_frag84 as base  # noqa: E402
from helpers import synthetic_fragment  # noqa: E402
from qemb_oracle import be, ccsd, ccsd_lambda, eri, scf  # noqa: E402


def main():
    t0 = time.time()
    N, O, NF, CEN = base.N, base.O, base.NF, base.CEN
    h, e1 = synthetic_fragment(N, O, base.SEED)
    h1, veff0, veff = base.energy_data(N, N)
    s4 = eri.pack_s4(e1)
    mf = scf.rhf(h, e1, O, conv_tol=1e-12, conv_tol_grad=1e-8)
    assert mf["converged"]
    C = mf["mo_coeff"]
    eris = ccsd.Eris(e1, C, O, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, nit = ccsd.kernel(eris, conv_tol=1e-12, conv_tol_normt=1e-10)
    assert conv
    print(f"RCCSD {ecc:.12f} in {nit} iterations ({time.time() - t0:.0f} s)", flush=True)
    z1, z2, nlam, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-11)
    print(f"Lambda in {nlam} iterations ({time.time() - t0:.0f} s)", flush=True)
    dm1, _ = ccsd_lambda.response_densities(lag, z1, z2)
    g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
    e_ref = be.get_frag_energy(C, O, NF, (1.0, CEN), np.zeros((N, N)), h1, dm1, g2, s4, veff0, None, True)
    np.savez(ROOT / "tests" / "golden" / "frag84_relaxed.npz", n=N, o=O, nf=NF, seed=base.SEED, cen=np.array(CEN), e_corr=ecc, n_iter=nit,
             lambda_iters=nlam, rdm1_emb=0.5 * C @ dm1 @ C.T, rdm1_mo=dm1, e_frag=np.array(e_ref), z1_norm=np.linalg.norm(z1), z2_norm=np.linalg.norm(z2))
    print("frag84_relaxed:", ecc, nlam, e_ref, f"({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
