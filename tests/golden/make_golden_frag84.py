"""Fixture for the -m gpu fragment test at the benchmark's n_occ = 20 (tests/test_gpu_fragment.py::test_fragment_at_bench_tiles).

The oracle (oracle/qemb_oracle: fragment RHF -> RCCSD -> unrelaxed RDMs -> get_frag_energy / update_ebe_hf) needs several
minutes at n = 84 on a handful of cores, so its outputs are stored here once and the GPU test compares the device against them.
n = 84, o = 20, v = 64: npair(o) = 210 and o(o-1)/2 = 190 packed pair rows with npair(v) = 2080 >= 2048 columns select the
224 x 128 / 192 x 128 ladder tiles (GEMM configs 23 / 25), o <= 32 the 128 x 32 / 32 x 128 tiles (20 / 21) -- the kernel
instantiations BASELINE configs[2] (o = 20, v = 200) runs on.

    python tests/golden/make_golden_frag84.py        (writes tests/golden/frag84.npz)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
from helpers import synthetic_fragment  # noqa: E402
from qemb_oracle import be, ccsd, eri, rdm, scf  # noqa: E402

N, O, NF, SEED = 84, 20, 30, 584
CEN = list(range(10))


def energy_data(n, seed):
    rng = np.random.default_rng(seed)
    mats = []
    for _ in range(3):
        a = rng.standard_normal((n, n)); mats.append(a + a.T)
    return mats


def main():
    h, e1 = synthetic_fragment(N, O, SEED)
    h1, veff0, veff = energy_data(N, N)
    s4 = eri.pack_s4(e1)
    mf = scf.rhf(h, e1, O, conv_tol=1e-12, conv_tol_grad=1e-8)
    assert mf["converged"]
    t1, t2, ecc, nit = ccsd.solve_ccsd(h, e1, O, mf["mo_coeff"], mf["mo_energy"], conv_tol=1e-11, conv_tol_normt=1e-9)
    C = mf["mo_coeff"]
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    r2 = rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
    e_ref = be.get_frag_energy(C, O, NF, (1.0, CEN), np.zeros((N, N)), h1, r1, r2, s4, veff0, None, True)
    f = be.Frag(list(range(NF)), 0, [], [], [], [], (1.0, CEN))
    f.h1, f.veff, f.TA, f._mo_coeffs, f.nsocc, f.eri_s4 = h1, veff, np.zeros((N, N)), C, O, s4
    np.savez(ROOT / "tests" / "golden" / "frag84.npz", n=N, o=O, nf=NF, seed=SEED, cen=np.array(CEN), e_scf=mf["e_tot"],
             mo_energy=mf["mo_energy"], e_corr=ecc, n_iter=nit, rdm1_emb=C @ r1 @ C.T * 0.5, e_frag=np.array(e_ref),
             ebe_hf=be.update_ebe_hf(f), t1_norm=np.linalg.norm(t1), t2_norm=np.linalg.norm(t2))
    print("frag84:", mf["e_tot"], ecc, nit, e_ref)


if __name__ == "__main__":
    main()
