"""Fixture for the -m gpu parity test AT THE BENCHMARKED SIZE (tests/test_gpu_fragment.py::test_bench_fragment_n220_against_oracle):
fragment 0 of bench.py's workload (BASELINE configs[2]: n = 220, n_occ = 20, n_virt = 200, seed 20260803, DF-factorised synthetic
ERIs with naux = 660, scale 0.03), solved by the oracle on the CPU.

The oracle runs its fragment RHF (oracle/qemb_oracle/scf.py) with J / K contracted from the DF factor the family is defined by (the
n^4 tensor is 18.7 GB and is not formed), assembles the MO blocks from the transformed factor (ccsd_lean.LeanEris) and runs the RCCSD
amplitude equations (ccsd_lean.update_amps == ccsd.update_amps, tests/test_oracle_ccsd.py): 3 plain updates from the MP2 guess (the
number bench.py's parity field compares) and the DIIS solve to |dE| < 1e-11.  About ten minutes on 8 cores, hence stored.

    python tests/golden/make_golden_frag220.py            (writes tests/golden/frag220.npz: fragment 0)
    python tests/golden/make_golden_frag220.py 1 2        (round 4: fragments 1 and 2 of the same sweep, seeds 20260804 / 05,
                                                           -> tests/golden/frag220_f1.npz, frag220_f2.npz)
    python tests/golden/make_golden_frag220.py 132:12     (round 5: the mid-size point of bench.py's size sweep, n = 132, n_occ = 12, same family and seed,
                                                           ERI scale 0.06 (55 / n)^(1/2) as tools/size_sweep.py -> tests/golden/frag132.npz; a minute)
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
from qemb_oracle import ccsd_lean, rdm, scf  # noqa: E402

N, O, SEED, SCALE = 220, 20, 20260803, 0.03


def bench_fragment(n, seed, scale):
    """the draws of bench.make_device_eris, in its order: DF factor, one-body part (the third draw, veff0, is not needed here)"""
    rng = np.random.default_rng(seed)
    naux = 3 * n
    B = scale * rng.standard_normal((naux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    A = rng.standard_normal((n, n))
    h = np.diag(2.0 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    return h, B


def main(frag=0, n=N, o=O, scale=SCALE):
    N, O, SCALE = n, o, scale          # (shadow the module constants: the body below is written in their terms)
    t0 = time.time()
    seed = SEED + frag
    out = ("frag220.npz" if frag == 0 else f"frag220_f{frag}.npz") if n == 220 else f"frag{n}.npz"
    h, B = bench_fragment(N, seed, SCALE)
    naux = B.shape[0]
    Bf = B.reshape(naux, -1)

    def jk(dm):
        J = (Bf.T @ (Bf @ dm.ravel())).reshape(N, N)
        X = B @ dm                                   # [P, p, s] = B[P,p,q] dm[q,s]
        K = np.einsum("Pps,Prs->pr", X, B, optimize=True)
        return J, K
    mf = scf.rhf(h, None, O, conv_tol=1e-12, conv_tol_grad=1e-8, jk=jk)
    assert mf["converged"]
    C = mf["mo_coeff"]
    print(f"RHF {mf['e_tot']:.12f} in {mf['cycles']} cycles ({time.time() - t0:.0f} s)", flush=True)
    B_mo = np.einsum("Ppq,pi,qj->Pij", B, C, C, optimize=True)
    er = ccsd_lean.LeanEris(B_mo, O, mf["mo_energy"])
    print(f"MO blocks ({time.time() - t0:.0f} s)", flush=True)
    t1, t2 = ccsd_lean.init_amps(er)
    e_mp2 = ccsd_lean.energy(t1, t2, er)
    for _ in range(3):
        t1, t2 = ccsd_lean.update_amps(t1, t2, er)
    e3 = ccsd_lean.energy(t1, t2, er)
    print(f"E(MP2) {e_mp2:.12f}  E(3 plain updates) {e3:.12f} ({time.time() - t0:.0f} s)", flush=True)
    conv, ecc, t1, t2, nit = ccsd_lean.kernel(er, conv_tol=1e-11, conv_tol_normt=1e-9)
    assert conv
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    np.savez_compressed(ROOT / "tests" / "golden" / out, n=N, o=O, seed=seed, scale=SCALE, e_scf=mf["e_tot"], mo_energy=mf["mo_energy"],
                        e_mp2=e_mp2, e_corr_3_plain_updates=e3, e_corr=ecc, n_iter=nit, rdm1_emb=C @ r1 @ C.T * 0.5,
                        t1_norm=np.linalg.norm(t1), t2_norm=np.linalg.norm(t2))
    print(f"{out}: E_corr {ecc:.12f} in {nit} iterations ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    for a in (sys.argv[1:] or ["0"]):
        if ":" in a:
            nn, oo = (int(x) for x in a.split(":"))
            main(0, nn, oo, 0.06 * min(1.0, (55.0 / nn) ** 0.5))
        else:
            main(int(a))
