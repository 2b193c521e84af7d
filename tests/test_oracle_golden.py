"""CPU: the oracle restatement against golden vectors produced by the reference's own NumPy functions
(tests/golden/make_golden.py).  This is what pins rows a1, a1', a9, a11, a12, a13, a15."""
import numpy as np

from helpers import GOLDEN, load_frag_lists
from qemb_oracle import be, eri, rdm, schmidt


def test_schmidt_eigh_matches_reference():
    g = np.load(GOLDEN / "schmidt.npz")
    for case in range(4):
        C, nocc, frag = g[f"C{case}"], int(g[f"nocc{case}"]), list(g[f"frag{case}"])
        TA, nf, nb = schmidt.schmidt_decomposition(C, nocc, frag)
        assert (nf, nb) == tuple(g[f"nfnb{case}"])
        ref = g[f"TA{case}"]
        assert TA.shape == ref.shape
        assert np.array_equal(TA[:, :nf], ref[:, :nf])
        # bath columns are eigenvectors: unique up to sign for non-degenerate eigenvalues
        for k in range(nf, nf + nb):
            s = np.sign(TA[:, k] @ ref[:, k])
            assert np.abs(TA[:, k] - s * ref[:, k]).max() < 1e-10


def test_schmidt_svd_spans_reference_bath():
    g = np.load(GOLDEN / "schmidt.npz")
    for case in range(4):
        C, nocc, frag = g[f"C{case}"], int(g[f"nocc{case}"]), list(g[f"frag{case}"])
        D = C[:, :nocc] @ C[:, :nocc].T
        TA = schmidt.schmidt_decomp_svd(D, frag)
        ref = g[f"TAsvd{case}"]
        assert TA.shape == ref.shape
        assert np.abs(TA @ TA.T - ref @ ref.T).max() < 1e-10
        # eigh- and svd-flavour baths span the same space for an idempotent D (SURVEY 8a a1')
        ref_e = g[f"TA{case}"]
        assert np.abs(TA @ TA.T - ref_e @ ref_e.T).max() < 1e-8


def test_rdms_match_reference():
    g = np.load(GOLDEN / "rdm.npz")
    for case in range(3):
        t1, t2 = g[f"t1_{case}"], g[f"t2_{case}"]
        assert np.array_equal(rdm.make_rdm1_ccsd_t1(t1), g[f"rdm1_{case}"])
        assert np.allclose(rdm.make_rdm2_urlx(t1, t2, True), g[f"rdm2_dm1_{case}"], atol=1e-15)
        assert np.allclose(rdm.make_rdm2_urlx(t1, t2, False), g[f"rdm2_cum_{case}"], atol=1e-15)


def _frags(tag, key, g):
    d = load_frag_lists(key)
    F = []
    for I in range(len(d["AO_per_frag"])):
        w, c = d["weight_and_relAO_per_center_per_frag"][I]
        f = be.Frag(d["AO_per_frag"][I], I, d["AO_per_edge_per_frag"][I], d["ref_frag_idx_per_edge_per_frag"][I],
                    d["relAO_per_edge_per_frag"][I], d["relAO_in_ref_per_edge_per_frag"][I], (w, c), d["relAO_per_origin_per_frag"][I])
        f._rdm1 = g[f"{tag}_rdm1_{I}"]
        f.h1 = np.zeros_like(f._rdm1)
        F.append(f)
    return d, F


def test_heff_and_solve_error_match_reference():
    g = np.load(GOLDEN / "be_pieces.npz")
    for tag, key, Nocc in [("h8", "test_autogen_h_linear_be2", 4), ("oct", "test_autogen_octane_be2", 33)]:
        d, F = _frags(tag, key, g)
        pot = be.initialize_pot(len(F), d["relAO_per_edge_per_frag"])
        assert len(pot) == int(g[f"{tag}_npot"])
        u = g[f"{tag}_u"]
        cout = 0
        for I, f in enumerate(F):
            f.udim = cout
            cout = be.set_udim(f, cout)
            assert f.udim == int(g[f"{tag}_udim{I}"])
            assert np.array_equal(be.update_heff(f, u), g[f"{tag}_heff{I}"])
            assert np.array_equal(be.update_heff(f, u, only_chem=True), g[f"{tag}_heffchem{I}"])
        nrm, vec = be.solve_error(F, Nocc)
        assert np.allclose(vec, g[f"{tag}_errvec"], atol=1e-15) and abs(nrm - float(g[f"{tag}_errnorm"])) < 1e-15
        nrm, vec = be.solve_error(F, Nocc, only_chem=True)
        assert np.allclose(vec, g[f"{tag}_errvec_chem"], atol=1e-14)


def test_fragment_energies_match_reference():
    g = np.load(GOLDEN / "be_pieces.npz")
    for case in range(2):
        meta = g[f"e{case}_meta"]
        n, o, nf = map(int, meta[:3]); cen = [int(x) for x in meta[3:]]
        mo, h1, veff0, veff = g[f"e{case}_mo"], g[f"e{case}_h1"], g[f"e{case}_veff0"], g[f"e{case}_veff"]
        t1, t2, eri4 = g[f"e{case}_t1"], g[f"e{case}_t2"], g[f"e{case}_eri4"]
        TA = np.zeros((n + 3, n))
        r1 = rdm.make_rdm1_ccsd_t1(t1)
        for cum in (True, False):
            r2 = rdm.make_rdm2_urlx(t1, t2, with_dm1=not cum)
            e = be.get_frag_energy(mo, o, nf, (1.0, cen), TA, h1, r1, r2, eri4, veff0, veff, cum)
            assert np.allclose(e, g[f"efrag{case}_{int(cum)}"], rtol=1e-12, atol=1e-12)
        f = be.Frag(list(range(nf)), 0, [], [], [], [], (1.0, cen))
        f.h1, f.veff, f.TA, f._mo_coeffs, f.nsocc, f.eri_s4 = h1, veff, TA, mo, o, eri4
        e_h1, e_coul, e_vec = be.update_ebe_hf(f, return_e=True)
        ref = g[f"ebehf{case}"]
        assert np.allclose([f.ebe_hf, e_h1, e_coul], ref, rtol=1e-12)
        assert np.allclose(e_vec, g[f"ebehf_vec{case}"], rtol=1e-12)


def test_pair_index_convention():
    g = np.load(GOLDEN / "misc.npz")
    r = g["ravel"]
    for a in range(12):
        for b in range(12):
            assert eri.ravel_symmetric(a, b) == r[a, b]
    # s4 <-> s1 round trip in that convention
    rng = np.random.default_rng(0)
    n = 7
    B = rng.standard_normal((5, n, n)); B = B + B.transpose(0, 2, 1)
    e1 = np.einsum("Ppq,Prs->pqrs", B, B)
    s4 = eri.pack_s4(e1)
    assert s4[eri.ravel_symmetric(4, 2), eri.ravel_symmetric(1, 6)] == e1[4, 2, 6, 1]
    assert np.array_equal(eri.restore_s1(s4, n), e1)
    assert np.array_equal(eri.restore_s1(eri.pack_s8(e1), n), e1)
