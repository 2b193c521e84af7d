"""Randomised parity sweep on the GPU: fragment pipeline (unrelaxed and relaxed) against the oracle for random (n, n_occ, n_f)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
from helpers import synthetic_fragment
from qemb_oracle import be, ccsd, ccsd_lambda, eri, rdm, scf
from quemb_amd.fragsolver import DeviceFragment, default_opts
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # sizes: n in [nmin, nmax) -- 60 .. 100 reaches the split-K slab paths (packed pair index >= 2048)
nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 41
worst = 0.0
t0 = time.time()
for case in range(ncase):
    n = int(rng.integers(nmin, nmax)); o = int(rng.integers(1, n)); nf = int(rng.integers(1, n + 1))
    cen = sorted(set(int(x) for x in rng.integers(0, nf, size=min(nf, 3))))
    h, e1 = synthetic_fragment(n, o, 1000 + case, scale=0.05)
    mf = scf.rhf(h, e1, o, conv_tol=1e-12, conv_tol_grad=1e-8)
    if not mf["converged"]:
        print("case", case, (n, o), "oracle SCF not converged: skipped"); continue
    h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
    v0 = rng.standard_normal((n, n)); v0 = v0 + v0.T
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf); fr.set_eri_s4(s4); fr.set_energy_data(h1, v0, None, 0.7, cen)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-12, conv_tol_normt=1e-10, max_cycle=200)
    if not conv:
        print("case", case, (n, o), "oracle CCSD not converged: skipped"); continue
    C = mf["mo_coeff"]
    for relax in (0, 1):
        out = fr.solve(o, h, opts=default_opts(relax_density=relax, cc_conv_tol=1e-12, cc_conv_tol_normt=1e-10, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8,
                                               lambda_conv_tol=1e-10, cc_max_cycle=200), eeval=True)
        if relax:
            z1, z2, _, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-11)
            dm1, _ = ccsd_lambda.response_densities(lag, z1, z2); g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
        else:
            dm1, g2 = rdm.make_rdm1_ccsd_t1(t1), rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
        e_ref = be.get_frag_energy(C, o, nf, (0.7, cen), np.zeros((n, n)), h1, dm1, g2, s4, v0, None, True)
        err = max(abs(out["e_corr_mo"] - ecc), np.abs(out["rdm1_emb"] - 0.5 * C @ dm1 @ C.T).max(), np.abs(np.array(out["e_frag"]) - np.array(e_ref)).max())
        worst = max(worst, err)
        if err > 1e-8:
            print("MISMATCH case", case, (n, o, nf, cen), "relax", relax, err, flush=True)
    fr.free()
print("fuzz done: %d cases, worst abs error %.2e, %.1f s" % (ncase, worst, time.time() - t0))
