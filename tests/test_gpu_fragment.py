"""-m gpu: the full per-fragment pipeline on the MI355X (through the C ABI) against the oracle.

Tolerances (north_star): fragment energies within 1e-8 Eh, 1-RDM within 1e-8 (both solvers converged
to |dE| < 1e-11, |dt| < 1e-9 so that the comparison is between fixed points)."""
import numpy as np
import pytest

from helpers import synthetic_fragment
from qemb_oracle import be, ccsd, eri, rdm, scf
from quemb_amd import eri_transform as et
from quemb_amd.fragsolver import DeviceFragment, default_opts

pytestmark = pytest.mark.gpu

TOL_E = 1e-8
TOL_RDM = 1e-8


def _energy_data(n, seed):
    rng = np.random.default_rng(seed)
    mats = []
    for _ in range(3):
        a = rng.standard_normal((n, n)); mats.append(a + a.T)
    return mats


@pytest.mark.parametrize("n,o,nf,cen", [(6, 2, 3, [0, 1]), (13, 5, 4, [0, 2]), (24, 6, 7, [1, 2, 3]), (42, 21, 21, list(range(7)))])
def test_fragment_pipeline_matches_oracle(qlib, n, o, nf, cen):
    h, e1 = synthetic_fragment(n, o, 500 + n)
    h1, veff0, veff = _energy_data(n, n)
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf)
    fr.set_eri_s4(s4)
    assert np.array_equal(fr.get_eri_s4(), s4)
    fr.set_energy_data(h1, veff0, veff, 1.0, cen)
    opts = default_opts(cc_conv_tol=1e-11, cc_conv_tol_normt=1e-9, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8)
    out = fr.solve(o, h, opts=opts, eeval=True)
    mf = scf.rhf(h, e1, o, conv_tol=1e-12, conv_tol_grad=1e-8)
    assert mf["converged"]
    t1, t2, ecc, nit = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"], conv_tol=1e-11, conv_tol_normt=1e-9)
    assert abs(out["e_scf"] - mf["e_tot"]) < TOL_E
    assert np.abs(out["mo_energy"] - mf["mo_energy"]).max() < 1e-7
    assert abs(out["e_corr_mo"] - ecc) < TOL_E, (out["e_corr_mo"], ecc)
    assert abs(out["n_iter"] - nit) <= 2
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    assert np.abs(out["rdm1_emb"] - mf["mo_coeff"] @ r1 @ mf["mo_coeff"].T * 0.5).max() < TOL_RDM
    r2 = rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
    e_ref = be.get_frag_energy(mf["mo_coeff"], o, nf, (1.0, cen), np.zeros((n, n)), h1, r1, r2, s4, veff0, None, True)
    assert np.abs(np.array(out["e_frag"]) - np.array(e_ref)).max() < TOL_E, (out["e_frag"], e_ref)
    f = be.Frag(list(range(nf)), 0, [], [], [], [], (1.0, cen))
    f.h1, f.veff, f.TA, f._mo_coeffs, f.nsocc, f.eri_s4 = h1, veff, np.zeros((n, n)), mf["mo_coeff"], o, s4
    assert abs(out["ebe_hf"] - be.update_ebe_hf(f)) < TOL_E


def test_one_call_abi_and_jk(qlib):
    import ctypes as C
    from quemb_amd._lib import check
    n, o, nf = 10, 4, 3
    h, e1 = synthetic_fragment(n, o, 77)
    s4 = eri.pack_s4(e1)
    h1, veff0, _ = _energy_data(n, 3)
    cen = np.array([0, 1], dtype=np.int32)
    opts = default_opts()
    mo = np.empty((n, n)); eps = np.empty(n); t1 = np.empty((o, n - o)); t2 = np.empty((o, o, n - o, n - o)); r = np.empty((n, n))
    ef = np.zeros(3); ec = C.c_double(); nit = C.c_int()
    check(qlib.qemb_ccsd_solve(n, o, nf, h.ctypes.data, s4.ctypes.data, None, C.byref(opts), h1.ctypes.data, veff0.ctypes.data, 1.0,
                               cen.ctypes.data_as(C.POINTER(C.c_int)), 2, mo.ctypes.data, eps.ctypes.data, t1.ctypes.data,
                               t2.ctypes.data, r.ctypes.data, ef.ctypes.data, C.byref(ec), C.byref(nit)))
    mf = scf.rhf(h, e1, o)
    t1o, t2o, ecc, _ = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"])
    assert abs(ec.value - ecc) < TOL_E
    # phase-invariant check of t2: the energy functional evaluated with the returned amplitudes and MOs
    eris = ccsd.Eris(e1, mo, o, mo_energy=eps)
    assert abs(ccsd.energy(t1, t2, eris) - ecc) < TOL_E
    fr = DeviceFragment(n, nf); fr.set_eri_s4(s4)
    P = np.random.default_rng(0).standard_normal((n, n)); P = P + P.T
    J, K = fr.jk(P)
    Jr, Kr = scf.get_jk(e1, P)
    assert np.abs(J - Jr).max() < 1e-11 and np.abs(K - Kr).max() < 1e-11


@pytest.mark.parametrize("n,o,nf,cen", [(6, 2, 3, [0, 1]), (13, 5, 4, [0, 2]), (24, 6, 7, [1, 2, 3])])
def test_relaxed_density_pipeline_matches_oracle(qlib, n, o, nf, cen):
    """relax_density=True (solve_ccsd(relax=True), molbe/solver.py:925-939): Lambda equations + response densities on the
    device against the oracle's reverse-mode restatement (pinned by energy derivatives, tests/test_oracle_lambda.py)."""
    from qemb_oracle import ccsd_lambda
    h, e1 = synthetic_fragment(n, o, 900 + n)
    h1, veff0, veff = _energy_data(n, n + 1)
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf)
    fr.set_eri_s4(s4)
    fr.set_energy_data(h1, veff0, veff, 1.0, cen)
    opts = default_opts(cc_conv_tol=1e-12, cc_conv_tol_normt=1e-10, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8, relax_density=1,
                        lambda_conv_tol=1e-10)
    out = fr.solve(o, h, opts=opts, eeval=True)
    mf = scf.rhf(h, e1, o, conv_tol=1e-12, conv_tol_grad=1e-8)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-12, conv_tol_normt=1e-10)
    z1, z2, nit, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-11)
    dm1, _ = ccsd_lambda.response_densities(lag, z1, z2)
    g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
    C = mf["mo_coeff"]
    assert abs(out["e_corr_mo"] - ecc) < TOL_E
    assert abs(out["lambda_iters"] - nit) <= 2
    assert np.abs(out["rdm1_emb"] - 0.5 * C @ dm1 @ C.T).max() < TOL_RDM
    e_ref = be.get_frag_energy(C, o, nf, (1.0, cen), np.zeros((n, n)), h1, dm1, g2, s4, veff0, None, True)
    assert np.abs(np.array(out["e_frag"]) - np.array(e_ref)).max() < TOL_E, (out["e_frag"], e_ref)


@pytest.mark.parametrize("n,o", [(2, 1), (3, 1), (3, 2), (4, 3), (5, 1), (5, 4)])
@pytest.mark.parametrize("relax", [0, 1])
def test_degenerate_fragment_sizes(qlib, n, o, relax):
    """One occupied or one virtual orbital: the pair-packed operands (ladder, tau-side dressing, MO transform) have empty
    antisymmetric blocks; amplitudes, energies and (relaxed) densities still match the oracle."""
    from qemb_oracle import ccsd_lambda
    h, e1 = synthetic_fragment(n, o, 300 + 10 * n + o, scale=0.12)
    fr = DeviceFragment(n, 1)
    fr.set_eri_s4(eri.pack_s4(e1))
    fr.set_energy_data(h, h, None, 1.0, [0])
    out = fr.solve(o, h, opts=default_opts(relax_density=relax, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11, lambda_conv_tol=1e-11), eeval=True)
    mf = scf.rhf(h, e1, o)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-13, conv_tol_normt=1e-11)
    assert abs(out["e_corr_mo"] - ecc) < 1e-11
    C = mf["mo_coeff"]
    if relax:
        z1, z2, _, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-12)
        dm1, _ = ccsd_lambda.response_densities(lag, z1, z2)
        g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
    else:
        dm1, g2 = rdm.make_rdm1_ccsd_t1(t1), rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
    assert np.abs(out["rdm1_emb"] - 0.5 * C @ dm1 @ C.T).max() < 1e-9
    e_ref = be.get_frag_energy(C, o, 1, (1.0, [0]), np.zeros((n, n)), h, dm1, g2, eri.pack_s4(e1), h, None, True)
    assert np.abs(np.array(out["e_frag"]) - np.array(e_ref)).max() < 1e-9


def test_fragment_at_bench_tiles(qlib):
    """n_occ = 20 (the benchmark's) with v = 64: the pp-ladder runs on the 224 x 128 / 192 x 128 tiles (GEMM configs 23 / 25: 210 and
    190 packed pair rows, npair(v) = 2080 >= 2048 columns) and the t1 contractions on the 128 x 32 / 32 x 128 tiles (20 / 21) -- the
    kernel instantiations the headline number is quoted on.  Expected values: the oracle's, stored once by
    tests/golden/make_golden_frag84.py (several minutes of NumPy)."""
    import sys
    from helpers import GOLDEN
    if str(GOLDEN) not in sys.path:
        sys.path.insert(0, str(GOLDEN))
    import make_golden_frag84 as mg
    g = np.load(GOLDEN / "frag84.npz")
    n, o, nf, cen = int(g["n"]), int(g["o"]), int(g["nf"]), [int(c) for c in g["cen"]]
    assert (n, o, nf, int(g["seed"])) == (mg.N, mg.O, mg.NF, mg.SEED)
    h, e1 = synthetic_fragment(n, o, int(g["seed"]))
    h1, veff0, veff = mg.energy_data(n, n)
    fr = DeviceFragment(n, nf)
    fr.set_eri_s4(eri.pack_s4(e1))
    fr.set_energy_data(h1, veff0, veff, 1.0, cen)
    opts = default_opts(cc_conv_tol=1e-11, cc_conv_tol_normt=1e-9, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8)
    out = fr.solve(o, h, opts=opts, eeval=True, want_t2=True)
    assert abs(out["e_scf"] - float(g["e_scf"])) < TOL_E
    assert np.abs(out["mo_energy"] - g["mo_energy"]).max() < 1e-7
    assert abs(out["e_corr_mo"] - float(g["e_corr"])) < TOL_E, (out["e_corr_mo"], float(g["e_corr"]))
    assert abs(out["n_iter"] - int(g["n_iter"])) <= 2
    assert np.abs(out["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
    assert np.abs(np.array(out["e_frag"]) - g["e_frag"]).max() < TOL_E, (out["e_frag"], g["e_frag"])
    assert abs(out["ebe_hf"] - float(g["ebe_hf"])) < TOL_E
    assert abs(np.linalg.norm(out["t2"]) - float(g["t2_norm"])) < 1e-7
    fr.free()


@pytest.mark.parametrize("frag", [0, 1, 2])
def test_bench_fragment_n220_against_oracle(qlib, frag):
    """(round 4: fragments 1 and 2 of the timed sweep as well, seeds 20260804 / 05 -> frag220_f1.npz, frag220_f2.npz)
    THE benchmarked fragment (BASELINE configs[2]: n = 220, n_occ = 20, n_virt = 200; fragment 0 of bench.py's sweep, ERIs built on the
    device from the DF factor exactly as bench.make_device_eris does) against the oracle's stored results (tests/golden/frag220.npz, written
    by make_golden_frag220.py: ten minutes of NumPy): fragment RHF, the energy after 3 plain amplitude updates from the MP2 guess (the number
    bench.py's parity field reports), and the converged DIIS solve -- E_corr, iteration count, 1-RDM."""
    import ctypes as C
    import sys
    from helpers import GOLDEN
    from quemb_amd._lib import DeviceBuffer, check
    if str(GOLDEN) not in sys.path:
        sys.path.insert(0, str(GOLDEN))
    import make_golden_frag220 as mg
    g = np.load(GOLDEN / ("frag220.npz" if frag == 0 else f"frag220_f{frag}.npz"))
    n, o = int(g["n"]), int(g["o"])
    assert (n, o, int(g["seed"]), float(g["scale"])) == (mg.N, mg.O, mg.SEED + frag, mg.SCALE) == (220, 20, 20260803 + frag, 0.03)
    h, B = mg.bench_fragment(n, int(g["seed"]), float(g["scale"]))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    npair = Bp.shape[1]
    dB, d4 = DeviceBuffer.from_numpy(Bp), DeviceBuffer(npair * npair)
    check(qlib.qemb_op_gemm(npair, npair, Bp.shape[0], 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
    fr = DeviceFragment(n, 22)
    fr.set_eri_s4_dev(d4.ptr); d4.free()
    fr.set_df_factor_dev(dB.ptr, Bp.shape[0])
    frd = DeviceFragment(n, 22)             # the same fragment living on its factor alone (round 5: no 4.7 GB block resident, J / K from the factor)
    frd.set_df_only_dev(dB.ptr, Bp.shape[0]); dB.free()
    assert frd.resident_bytes() == 8 * Bp.size
    fr.set_mo_route(0)                      # first the four quarter transformations of the packed block (what PySCF's ao2mo does) ...
    opts = default_opts(cc_conv_tol=1e-11, cc_conv_tol_normt=1e-9, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8)
    r = fr.scf(o, h, None, opts=opts)
    assert abs(r["e_scf"] - float(g["e_scf"])) < 1e-8 * max(1.0, abs(float(g["e_scf"])))
    assert np.abs(r["mo_energy"] - g["mo_energy"]).max() < 1e-8
    dm0 = 2.0 * r["mo_coeff"][:, :o] @ r["mo_coeff"][:, :o].T
    # (i) plain updates, no DIIS: the same arithmetic in the same order of iterations as the oracle's loop
    fr.prepare_ccsd(o, h, dm0, opts=opts)
    e3, _ = fr.ccsd_iterate(3)
    assert abs(e3 - float(g["e_corr_3_plain_updates"])) < 1e-9, (e3, float(g["e_corr_3_plain_updates"]))
    # (ii) the product solve to convergence
    out = fr.solve(o, h, dm0, opts=opts, eeval=False, want_t2=True)
    assert abs(out["e_corr_mo"] - float(g["e_corr"])) < TOL_E, (out["e_corr_mo"], float(g["e_corr"]))
    assert abs(out["n_iter"] - int(g["n_iter"])) <= 1
    assert np.abs(out["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
    assert abs(np.linalg.norm(out["t1"]) - float(g["t1_norm"])) < 1e-7 and abs(np.linalg.norm(out["t2"]) - float(g["t2_norm"])) < 1e-7
    assert fr.mo_route_used() == (False, 3 * n)
    # ... then the route the bench takes (naux = 3 n: by cost): MO integrals from the fragment's 3-index factor, against the same oracle results
    fr.set_mo_route(-1)
    fr.prepare_ccsd(o, h, dm0, opts=opts)
    e3f, _ = fr.ccsd_iterate(3)
    assert abs(e3f - float(g["e_corr_3_plain_updates"])) < 1e-9, (e3f, float(g["e_corr_3_plain_updates"]))
    outf = fr.solve(o, h, dm0, opts=opts, eeval=False, want_t2=True)
    assert fr.mo_route_used() == (True, 3 * n)
    assert abs(outf["e_corr_mo"] - float(g["e_corr"])) < TOL_E, (outf["e_corr_mo"], float(g["e_corr"]))
    assert abs(outf["e_corr_mo"] - out["e_corr_mo"]) < 1e-10 and outf["n_iter"] == out["n_iter"]
    assert np.abs(outf["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
    assert np.abs(outf["t2"] - out["t2"]).max() < 1e-10 and np.abs(outf["t1"] - out["t1"]).max() < 1e-10
    # ... and the fragment that holds nothing but the factor: its own fragment RHF (J / K from the factor), the plain updates and the converged solve
    rd = frd.scf(o, h, None, opts=opts)
    assert abs(rd["e_scf"] - float(g["e_scf"])) < 1e-8 * max(1.0, abs(float(g["e_scf"])))
    assert np.abs(rd["mo_energy"] - g["mo_energy"]).max() < 1e-8
    assert np.abs(rd["J"] - r["J"]).max() < 1e-10 and np.abs(rd["K"] - r["K"]).max() < 1e-10
    frd.prepare_ccsd(o, h, dm0, opts=opts)
    e3d, _ = frd.ccsd_iterate(3)
    assert abs(e3d - float(g["e_corr_3_plain_updates"])) < 1e-9, (e3d, float(g["e_corr_3_plain_updates"]))
    outd = frd.solve(o, h, dm0, opts=opts, eeval=False, want_t2=False)
    assert frd.mo_route_used() == (True, 3 * n)
    assert abs(outd["e_corr_mo"] - float(g["e_corr"])) < TOL_E and abs(outd["n_iter"] - int(g["n_iter"])) <= 1
    assert np.abs(outd["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
    assert frd.resident_bytes() < 8 * Bp.size + 64 * n * n + 8 * (o * (n - o) + (o * (n - o)) ** 2) + 4096      # factor + orbitals / densities + kept amplitudes: no block
    frd.free()
    if frag == 0:
        # the fragment energies of the timed sweep (eeval: the 3/4-transformed integrals come from ONE more product on the factor route) by both routes
        rng = np.random.default_rng(7)
        V = rng.standard_normal((n, n))
        fr.set_energy_data(h, 0.05 * (V + V.T), None, 1.0, list(range(4, 8)))
        ef = fr.solve(o, h, dm0, opts=opts, eeval=True)
        fr.set_mo_route(0)
        e4 = fr.solve(o, h, dm0, opts=opts, eeval=True)
        assert np.abs(ef["e_frag"] - e4["e_frag"]).max() < 1e-9 and abs(ef["ebe_hf"] - e4["ebe_hf"]) < 1e-9, (ef["e_frag"], e4["e_frag"])
    fr.free()


def test_mid_size_fragment_n132_against_oracle(qlib):
    """The mid-size point of bench.py's size sweep (n = 132, n_occ = 12: rings on 64 x 64 tiles, ladder on the 80 x 128 tile of round 5, graph-replayed update)
    against the oracle's stored results (tests/golden/frag132.npz, make_golden_frag220.py 132:12): fragment RHF, three plain updates, the converged solve --
    with the fragment living on its factor and with the 4-fold block resident (four-index route)."""
    import sys
    from helpers import GOLDEN
    from quemb_amd._lib import DeviceBuffer, check
    if str(GOLDEN) not in sys.path:
        sys.path.insert(0, str(GOLDEN))
    import make_golden_frag220 as mg
    g = np.load(GOLDEN / "frag132.npz")
    n, o = int(g["n"]), int(g["o"])
    assert (n, o) == (132, 12)
    h, B = mg.bench_fragment(n, int(g["seed"]), float(g["scale"]))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    opts = default_opts(cc_conv_tol=1e-11, cc_conv_tol_normt=1e-9, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8)
    for mode in ("factor", "block"):
        fr = DeviceFragment(n, 22)
        if mode == "factor":
            fr.set_df_only(Bp)
        else:
            fr.set_eri_s4(Bp.T @ Bp); fr.set_mo_route(0)
        r = fr.scf(o, h, None, opts=opts)
        assert abs(r["e_scf"] - float(g["e_scf"])) < 1e-8 * max(1.0, abs(float(g["e_scf"])))
        assert np.abs(r["mo_energy"] - g["mo_energy"]).max() < 1e-8
        dm0 = 2.0 * r["mo_coeff"][:, :o] @ r["mo_coeff"][:, :o].T
        fr.prepare_ccsd(o, h, dm0, opts=opts)
        e3, _ = fr.ccsd_iterate(3)
        assert abs(e3 - float(g["e_corr_3_plain_updates"])) < 1e-9, (mode, e3, float(g["e_corr_3_plain_updates"]))
        out = fr.solve(o, h, dm0, opts=opts, eeval=False, want_t2=True)
        assert fr.mo_route_used()[0] == (mode == "factor")
        assert abs(out["e_corr_mo"] - float(g["e_corr"])) < TOL_E, (mode, out["e_corr_mo"], float(g["e_corr"]))
        assert abs(out["n_iter"] - int(g["n_iter"])) <= 1
        assert np.abs(out["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
        assert abs(np.linalg.norm(out["t1"]) - float(g["t1_norm"])) < 1e-7 and abs(np.linalg.norm(out["t2"]) - float(g["t2_norm"])) < 1e-7
        fr.free()


def test_factor_route_equals_four_index(qlib):
    """the 3-index factor route of the MO integrals on the HIP library: the checker of the host-logic suite at sizes that reach the tiled kernels
    (24 ... 96 orbitals; relaxed densities, energies with eeval, CPHF), then the DF transform's hand-over of its factor to the fragment"""
    from test_hostlogic_fragment import check_factor_route_equals_four_index
    check_factor_route_equals_four_index(qlib, cases=((8, 3, 3, 20), (24, 7, 6, 0), (45, 12, 10, 100), (33, 33, 5, 0), (96, 20, 22, 0)), tol=5e-10)


def test_fragment_living_on_its_factor(qlib):
    """qemb_frag_set_df_only on the device (J / K from the factor through the MFMA products, no resident block), incl. sizes on the large tiles"""
    from test_hostlogic_fragment import check_fragment_living_on_its_factor
    check_fragment_living_on_its_factor(qlib, cases=((8, 3, 3, 20), (24, 7, 6, 0), (45, 12, 10, 100), (33, 33, 5, 0), (96, 20, 22, 0)), tol=5e-10, jk_tol=1e-11)


def test_relaxed_fragment_at_bench_tiles(qlib):
    """relax_density = 1 at n_occ = 20, n_virt = 64 (the kernel instantiations of the headline workload): Lambda equations, response 1-RDM and
    the fragment energies from the relaxed 2-RDM against the oracle's stored results (tests/golden/frag84_relaxed.npz, written by
    make_golden_frag84_relaxed.py: minutes of NumPy).  solver.py:925-939."""
    import sys
    from helpers import GOLDEN
    if str(GOLDEN) not in sys.path:
        sys.path.insert(0, str(GOLDEN))
    import make_golden_frag84 as mg
    g = np.load(GOLDEN / "frag84_relaxed.npz")
    n, o, nf, cen = int(g["n"]), int(g["o"]), int(g["nf"]), [int(c) for c in g["cen"]]
    assert (n, o, nf, int(g["seed"])) == (mg.N, mg.O, mg.NF, mg.SEED)
    h, e1 = synthetic_fragment(n, o, int(g["seed"]))
    h1, veff0, veff = mg.energy_data(n, n)
    fr = DeviceFragment(n, nf)
    fr.set_eri_s4(eri.pack_s4(e1))
    fr.set_energy_data(h1, veff0, veff, 1.0, cen)
    opts = default_opts(cc_conv_tol=1e-12, cc_conv_tol_normt=1e-10, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8, relax_density=1, lambda_conv_tol=1e-10)
    out = fr.solve(o, h, opts=opts, eeval=True)
    assert abs(out["e_corr_mo"] - float(g["e_corr"])) < TOL_E, (out["e_corr_mo"], float(g["e_corr"]))
    assert abs(out["lambda_iters"] - int(g["lambda_iters"])) <= 2
    assert np.abs(out["rdm1_emb"] - g["rdm1_emb"]).max() < TOL_RDM
    assert np.abs(np.array(out["e_frag"]) - g["e_frag"]).max() < TOL_E, (out["e_frag"], g["e_frag"])
    fr.free()


# randomised parity (round 4: tools/fuzz_parity.py as a test): fixed seeds, sizes 6 .. 96, random n_occ, fragment sites and centres;
# fragment RHF -> RCCSD -> densities -> fragment energies against the oracle, unrelaxed everywhere and relaxed up to n = 48
FUZZ_CASES = [(seed, 6, 41) for seed in range(101, 113)] + [(201, 41, 61), (202, 41, 61), (203, 61, 81), (204, 61, 81), (205, 81, 97), (206, 96, 97)]


@pytest.mark.parametrize("seed,nmin,nmax", FUZZ_CASES)
def test_randomised_fragment_parity(qlib, seed, nmin, nmax):
    from qemb_oracle import ccsd_lambda
    rng = np.random.default_rng(seed)
    n = int(rng.integers(nmin, nmax))
    o = int(rng.integers(1, n)) if n <= 40 else int(rng.integers(4, max(5, n // 4)))      # large cases: the benchmark's regime n_occ << n_virt (oracle cost)
    nf = int(rng.integers(1, n + 1))
    cen = sorted(set(int(x) for x in rng.integers(0, nf, size=min(nf, 3))))
    h, e1 = synthetic_fragment(n, o, 1000 + seed, scale=0.05 if n <= 40 else None)
    mf = scf.rhf(h, e1, o, conv_tol=1e-12, conv_tol_grad=1e-8)
    assert mf["converged"]
    h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
    v0 = rng.standard_normal((n, n)); v0 = v0 + v0.T
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf); fr.set_eri_s4(s4); fr.set_energy_data(h1, v0, None, 0.7, cen)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-12, conv_tol_normt=1e-10, max_cycle=200)
    assert conv
    C = mf["mo_coeff"]
    for relax in ((0, 1) if n <= 48 else (0,)):
        out = fr.solve(o, h, opts=default_opts(relax_density=relax, cc_conv_tol=1e-12, cc_conv_tol_normt=1e-10, scf_conv_tol=1e-12, scf_conv_tol_grad=1e-8,
                                               lambda_conv_tol=1e-10, cc_max_cycle=200), eeval=True)
        if relax:
            z1, z2, _, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-11)
            dm1, _ = ccsd_lambda.response_densities(lag, z1, z2); g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
        else:
            dm1, g2 = rdm.make_rdm1_ccsd_t1(t1), rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
        e_ref = be.get_frag_energy(C, o, nf, (0.7, cen), np.zeros((n, n)), h1, dm1, g2, s4, v0, None, True)
        assert abs(out["e_corr_mo"] - ecc) < TOL_E, (n, o, relax, out["e_corr_mo"], ecc)
        assert np.abs(out["rdm1_emb"] - 0.5 * C @ dm1 @ C.T).max() < TOL_RDM, (n, o, relax)
        assert np.abs(np.array(out["e_frag"]) - np.array(e_ref)).max() < TOL_E, (n, o, nf, cen, relax)
    fr.free()


def test_wide_diis_space_takes_the_general_path(qlib):
    """cc_diis_space = 10 > 8: the pass-by-pass end of the iteration on the device (the fused launches take up to eight stored vectors)."""
    from test_hostlogic_fragment import check_wide_diis_space_takes_the_general_path
    check_wide_diis_space_takes_the_general_path(qlib)


def test_fragment_without_virtual_orbitals(qlib):
    from test_hostlogic_fragment import check_fragment_without_virtual_orbitals
    check_fragment_without_virtual_orbitals(qlib)


def test_non_strict_convergence_returns_results_with_a_warning(qlib):
    from test_hostlogic_fragment import check_non_strict_convergence
    check_non_strict_convergence(qlib)


@pytest.mark.parametrize("n,o", [(200, 80), (222, 150)])
def test_mo_transform_at_bench_tile(qlib, n, o):
    """192 < n <= 224: mo_transform (ccsd.cpp:48) runs its quarter transforms on the 224 x 128 tile (tcfg = 13), as at the benchmark's
    n = 220.  The MO blocks exported from the device are compared with blocks assembled on the host from the density-fitting factor
    of the synthetic fragment, (pq|rs) = sum_P B[P,p,q] B[P,r,s], rotated with the DEVICE's own orbitals (so that orbital phases and
    rotations inside degenerate shells cannot enter)."""
    from quemb_amd._lib import DeviceBuffer, check
    rng = np.random.default_rng(n + o)
    naux, v = 64, n - o
    B = 0.03 * rng.standard_normal((naux, n, n)); B = 0.5 * (B + B.transpose(0, 2, 1))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    npair = Bp.shape[1]
    dB, d4 = DeviceBuffer.from_numpy(Bp), DeviceBuffer(npair * npair)
    check(qlib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
    A = rng.standard_normal((n, n))
    h = np.diag(0.5 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)        # small gap: the orbitals mix strongly
    fr = DeviceFragment(n, 8)
    fr.set_eri_s4_dev(d4.ptr)
    dB.free(); d4.free()
    r = fr.scf(o, h, None)
    assert r["converged"]
    C = r["mo_coeff"]
    fr.prepare_ccsd(o, h, 2.0 * C[:, :o] @ C[:, :o].T)
    C = fr.ccsd_export("mo_coeff", (n, n))                # prepare_ccsd re-runs the fragment RHF from dm0: take ITS orbitals
    Bm = np.einsum("Ppq,pi,qj->Pij", B, C, C, optimize=True)
    f = lambda x: np.ascontiguousarray(x).reshape(naux, -1)
    Boo, Bov, Bvv = Bm[:, :o, :o], Bm[:, :o, o:], Bm[:, o:, o:]
    tol = 1e-11
    assert np.abs(fr.ccsd_export("oooo", (o, o, o, o)).reshape(o * o, -1) - f(Boo).T @ f(Boo)).max() < tol
    assert np.abs(fr.ccsd_export("ovoo", (o, v, o, o)).reshape(o * v, -1) - f(Bov).T @ f(Boo)).max() < tol
    assert np.abs(fr.ccsd_export("ovov", (o, v, o, v)).reshape(o * v, -1) - f(Bov).T @ f(Bov)).max() < tol
    assert np.abs(fr.ccsd_export("ovvv", (o, v, v, v)).reshape(o * v, -1) - f(Bov).T @ f(Bvv)).max() < tol
    assert np.abs(fr.ccsd_export("W1base", (o, v, o, v)) - (f(Bov).T @ f(Bov.transpose(0, 2, 1))).reshape(o, v, v, o).transpose(3, 2, 0, 1)).max() < tol
    assert np.abs(fr.ccsd_export("W2base", (o, v, o, v)) - (f(Boo).T @ f(Bvv)).reshape(o, o, v, v).transpose(1, 2, 0, 3)).max() < tol
    # (+/-) pair-packed ladder operands: Vp[P(ab),P(cd)] = (ac|bd) + (ad|bc), Vm[Q(ab),Q(cd)] = (ac|bd) - (ad|bc)
    npv, nmv = v * (v + 1) // 2, v * (v - 1) // 2
    ldp, ldm = npv + (npv & 1), max(nmv + (nmv & 1), 2)
    Vp = fr.ccsd_export("Vp", (npv, ldp)); Vm = fr.ccsd_export("Vm", (max(nmv, 1), ldm))
    vv = (f(Bvv).T @ f(Bvv)).reshape(v, v, v, v)             # (ab|cd) at [a,b,c,d]
    Vac = vv.transpose(0, 2, 1, 3)                           # [a,b,c,d] = (ac|bd)
    ilv, slv = np.tril_indices(v), np.tril_indices(v, -1)
    assert np.abs(Vp[:, :npv] - (Vac + Vac.transpose(0, 1, 3, 2))[ilv[0], ilv[1]][:, ilv[0], ilv[1]]).max() < tol
    assert np.abs(Vm[:, :nmv] - (Vac - Vac.transpose(0, 1, 3, 2))[slv[0], slv[1]][:, slv[0], slv[1]]).max() < tol
    assert np.abs(fr.ccsd_export("Vl", (v, v, v, v)) - Vac).max() < tol
    eo, ev = fr.ccsd_export("eo", (o,)), fr.ccsd_export("ev", (v,))
    assert np.abs(np.concatenate([eo, ev]) - r["mo_energy"]).max() < 1e-9
    fr.free()


def test_full_size_fragment_is_invariant_under_a_rotation_of_the_embedding_basis(qlib):
    """BASELINE configs[2] size (n = 220, n_occ = 20), checked through a size-independent property: rotating the embedding orbitals by
    an orthogonal Q -- the ERIs through the dense transform of row a3 (qemb_ao2mo_dense with TA = Q, N = n = 220), h -> Q^T h Q --
    leaves the fragment RHF energy and E_corr unchanged and turns the 1-RDM into Q^T rdm1 Q.  The whole chain a3 -> a7 -> a8 -> a9
    at full size, two independent solves (different orbitals, different integrals in memory)."""
    from quemb_amd._lib import DeviceBuffer, check
    n, o, naux = 220, 20, 330
    rng = np.random.default_rng(20260803)
    B = 0.03 * rng.standard_normal((naux, n, n)); B = 0.5 * (B + B.transpose(0, 2, 1))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    npair = Bp.shape[1]
    dB, d4 = DeviceBuffer.from_numpy(Bp), DeviceBuffer(npair * npair)
    check(qlib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
    dB.free()
    A = rng.standard_normal((n, n))
    h = np.diag(2.0 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    fr = DeviceFragment(n, 22)
    fr.set_eri_s4_dev(d4.ptr); d4.free()
    out = fr.solve(o, h, opts=default_opts(), eeval=False)
    eri = fr.get_eri_s4()
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ao = et.AOEri(eri, n, lib=qlib)
    del eri
    fr2 = DeviceFragment(n, 22)
    ao.transform(Q, frag=fr2, want_host=False)
    ao.free()
    out2 = fr2.solve(o, Q.T @ h @ Q, opts=default_opts(), eeval=False)
    assert out["n_iter"] > 5 and abs(out["e_corr_mo"]) > 1.0          # a correlated, converged problem
    assert abs(out2["e_scf"] - out["e_scf"]) < 1e-9 * abs(out["e_scf"]), (out["e_scf"], out2["e_scf"])
    assert abs(out2["e_corr_mo"] - out["e_corr_mo"]) < 1e-8, (out["e_corr_mo"], out2["e_corr_mo"])
    assert np.abs(out2["rdm1_emb"] - Q.T @ out["rdm1_emb"] @ Q).max() < 1e-8
    # a spot check of the rotated integrals themselves against the DF factor: (00|00), (n-1 0|5 3)
    Bq = np.einsum("Ppq,pi,qj->Pij", B, Q, Q, optimize=True)
    e2 = fr2.get_eri_s4()
    pr = lambda i, j: i * (i + 1) // 2 + j
    for (i, j, k, l) in ((0, 0, 0, 0), (n - 1, 0, 5, 3), (100, 37, 219, 218)):
        assert abs(e2[pr(i, j), pr(k, l)] - Bq[:, i, j] @ Bq[:, k, l]) < 1e-11
    fr.free(); fr2.free()


def test_fragment_larger_than_the_benchmark_size(qlib):
    """n = 300, n_occ = 30 (n_virt = 270): past every tile configuration tuned for the benchmark -- 465 packed pair rows (no single-tile
    ladder), n > 224 (two row tiles in the MO transformation), ov = 8100 ring products -- checked through the size-independent property of
    the full-size test: an orthogonal rotation of the embedding basis (here applied to the density-fitting factor the ERIs are built
    from) leaves the fragment RHF energy and E_corr unchanged and rotates the 1-RDM."""
    from quemb_amd._lib import DeviceBuffer, check
    n, o, naux = 300, 30, 240
    rng = np.random.default_rng(300)
    B = 0.022 * rng.standard_normal((naux, n, n)); B = 0.5 * (B + B.transpose(0, 2, 1))
    A = rng.standard_normal((n, n))
    h = np.diag(2.0 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    il = np.tril_indices(n)
    npair = len(il[0])
    outs = []
    for Bx, hx in ((B, h), (np.einsum("Ppq,pi,qj->Pij", B, Q, Q, optimize=True), Q.T @ h @ Q)):
        Bp = np.ascontiguousarray(Bx[:, il[0], il[1]])
        dB, d4 = DeviceBuffer.from_numpy(Bp), DeviceBuffer(npair * npair)
        check(qlib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
        dB.free()
        fr = DeviceFragment(n, 22)
        fr.set_eri_s4_dev(d4.ptr); d4.free()
        outs.append(fr.solve(o, hx, opts=default_opts(), eeval=False))
        fr.free()
        check(qlib.qemb_trim())
    a, b = outs
    assert a["n_iter"] > 5 and abs(a["e_corr_mo"]) > 1.0
    assert abs(a["e_scf"] - b["e_scf"]) < 1e-9 * abs(a["e_scf"])
    assert abs(a["e_corr_mo"] - b["e_corr_mo"]) < 1e-8, (a["e_corr_mo"], b["e_corr_mo"])
    assert np.abs(b["rdm1_emb"] - Q.T @ a["rdm1_emb"] @ Q).max() < 1e-8


def test_solve_batch_lockstep_equals_one_by_one(qlib):
    """qemb_frag_solve_batch on the HIP library: grouped launches of the lock-step CCSD iterations, results identical bit for bit"""
    from test_hostlogic_fragment import check_solve_batch_equals_one_by_one
    check_solve_batch_equals_one_by_one(qlib, sizes=((12, 4, 4), (16, 5, 5), (10, 3, 3), (14, 6, 4), (9, 9, 2), (20, 7, 6)), expect_grouped=True)
