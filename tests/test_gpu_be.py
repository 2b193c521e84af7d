"""-m gpu: ERI transforms, Schmidt decomposition and the whole BE driver on the MI355X against the oracle and the
reference's end-to-end golden energies (H8 and octane, STO-3G)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
from qemb_oracle import eri as oeri
from quemb_amd import _lib, eri_transform as et

pytestmark = pytest.mark.gpu


def test_dense_transform_matches_oracle(qlib):
    rng = np.random.default_rng(50)
    N, n = 26, 11
    Bm = rng.standard_normal((40, N, N)); Bm = Bm + Bm.transpose(0, 2, 1)
    e1 = np.einsum("Ppq,Prs->pqrs", Bm, Bm)
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n]
    ref = oeri.ao2mo_full(e1, TA)
    for arr in (e1, oeri.pack_s4(e1), oeri.pack_s8(e1)):
        ao = et.AOEri(arr, N)
        got = ao.transform(TA)
        assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()


def test_df_transform_matches_oracle_and_dense(qlib):
    rng = np.random.default_rng(51)
    N, n, naux = 20, 9, 260
    L = rng.standard_normal((naux, N, N)); L = L + L.transpose(0, 2, 1)
    A = rng.standard_normal((naux, naux)); j2c = A @ A.T + naux * np.eye(naux)
    pqL = np.ascontiguousarray(L.transpose(1, 2, 0))
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n]
    ref = oeri.integral_direct_DF(pqL, j2c, TA)
    il = np.tril_indices(N)
    for layout, ints in [("pqL", pqL), ("Lpq", L), ("packed", np.ascontiguousarray(L[:, il[0], il[1]]))]:
        df = et.DFContext(j2c=j2c); df.set_ints(ints, N, layout)
        assert np.abs(df.transform(TA) - ref).max() < 1e-10 * np.abs(ref).max()
    df = et.DFContext(L_PQ=np.linalg.cholesky(j2c)); df.set_ints(L, N, "Lpq")
    assert np.abs(df.transform(TA) - ref).max() < 1e-10 * np.abs(ref).max()
    with pytest.raises(Exception):
        et.DFContext(j2c=-np.eye(4))          # not positive definite -> error, like scipy.linalg.cholesky


def test_schmidt_matches_reference_goldens(qlib):
    g = np.load(GOLDEN / "schmidt.npz")
    for case in range(4):
        Cm, nocc, frag = g[f"C{case}"], int(g[f"nocc{case}"]), list(g[f"frag{case}"])
        TA, nf, nb = et.schmidt_decomposition(Cm, nocc, frag)
        TAs_, nfs_, nbs_ = et.schmidt_decomposition(Cm, nocc, frag, lib=None, method="subspace")
        assert (nfs_, nbs_) == (nf, nb) and np.abs(TAs_ @ TAs_.T - TA @ TA.T).max() < 1e-9
        for k in range(nf, nf + nb):      # same eigenvectors (up to sign), same order
            assert min(np.abs(TAs_[:, k] - TA[:, k]).max(), np.abs(TAs_[:, k] + TA[:, k]).max()) < 1e-7
        ref = g[f"TA{case}"]
        assert (nf, nb) == tuple(g[f"nfnb{case}"]) and TA.shape == ref.shape
        assert np.array_equal(TA[:, :nf], ref[:, :nf])
        assert np.abs(TA @ TA.T - ref @ ref.T).max() < 1e-10
        assert np.abs(TA.T @ TA - np.eye(nf + nb)).max() < 1e-11
        D = Cm[:, :nocc] @ Cm[:, :nocc].T
        TAs = et.schmidt_decomp_svd(D, frag)
        assert np.abs(TAs @ TAs.T - g[f"TAsvd{case}"] @ g[f"TAsvd{case}"].T).max() < 1e-9


def test_schmidt_large_environment(qlib):
    """N_lo = 600: eigh of a 578 x 578 projector block by Jacobi sweeps; compared with LAPACK through projectors."""
    rng = np.random.default_rng(52)
    N, nocc = 600, 150
    frag = list(range(100, 122))
    Cm = np.linalg.qr(rng.standard_normal((N, N)))[0]
    TA, nf, nb = et.schmidt_decomposition(Cm, nocc, frag)
    from qemb_oracle import schmidt as os_
    ref, _, nbr = os_.schmidt_decomposition(Cm, nocc, frag)
    assert nb == nbr == 22
    assert np.abs(TA @ TA.T - ref @ ref.T).max() < 1e-9
    D = Cm[:, :nocc] @ Cm[:, :nocc].T
    assert abs(np.trace(TA.T @ D @ TA) - round(np.trace(TA.T @ D @ TA))) < 1e-9     # integer electron pairs in the embedding


def _be(which, **kw):
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    if which == "h8":
        mol = Mole([["H", (0.0, 0.0, float(i))] for i in range(8)]); key = "test_autogen_h_linear_be2"
    else:
        mol = Mole(GOLDEN / "octane.xyz"); key = "test_autogen_octane_be2"
    mf = RHF(mol); mf.kernel()
    return mf, BE(mf, FragPart.from_json(GOLDEN / "fragmentation.json", key), distribute=False, **kw)


def test_h8_oneshot_golden(qlib):
    mf, be = _be("h8")
    assert abs(be.hf_err) < 1e-8
    e, _ = be.oneshot()
    assert abs(e - (-0.13198886164212092)) < 3e-7      # tests/_expected_data_for_fragmentation_test.py:983 (PySCF conv_tol 1e-7)


def test_octane_hf_oneshot_and_density_matching_goldens(qlib):
    mf, be = _be("octane")
    assert abs(mf.e_tot - (-309.7847696458918)) < 5e-8          # tests/molbe_octane_test.py:32-36 (E_HF)
    assert abs(be.ebe_hf - (-309.7847696458918)) < 5e-8         # HF-in-HF
    e, _ = be.oneshot()
    assert abs(e - (-0.5499456086311243)) < 5e-7                # tests/_expected_data_for_fragmentation_test.py:984
    opt = be.optimize(solver="CCSD", only_chem=False)           # reference defaults: conv_tol 1e-6, QN, HF Jacobian
    assert opt.err < 1e-6
    # tests/molbe_octane_test.py:32-36: E_corr = -0.5499514850769742, E_tot = -310.3347211309688 (np.isclose, rtol 1e-5)
    assert abs(be.e_corr - (-0.5499514850769742)) < 5e-6
    assert abs(be.ebe_tot - (-310.3347211309688)) < 5e-6


def test_h8_df_transform_path(qlib):
    """int-direct-DF-hip with a (numerically) complete auxiliary set reproduces the in-core result."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    mol = Mole([["H", (0.0, 0.0, float(i))] for i in range(8)])
    mf = RHF(mol); mf.kernel()
    N = mol.nao
    e1 = mf._eri
    il = np.tril_indices(N)
    j2c = oeri.pack_s4(e1) + 1e-10 * np.eye(N * (N + 1) // 2)      # AO pairs as the auxiliary set
    pqL = np.zeros((N, N, j2c.shape[0])); pqL[il[0], il[1], :] = oeri.pack_s4(e1); pqL[il[1], il[0], :] = oeri.pack_s4(e1)
    fobj = FragPart.from_json(GOLDEN / "fragmentation.json", "test_autogen_h_linear_be2")
    be_df = BE(mf, fobj, int_transform="int-direct-DF-hip", df_ints=(pqL, j2c, "pqL"), distribute=False)
    be_in = BE(mf, fobj, distribute=False)
    e_in = be_in.oneshot()[0]
    e_df = be_df.oneshot()[0]
    assert abs(e_df - e_in) < 1e-6
    # the DF transform left its fitted factor with every fragment, and the sweep formed the MO integrals from it (36 auxiliary functions <= 8 n);
    # the four-index transformation of the same fragments gives the same energy to rounding, the in-core fragments have no factor
    assert all(f.dev.mo_route_used() == (True, j2c.shape[0]) for f in be_df.Fobjs)
    assert all(f.dev.mo_route_used() == (False, 0) for f in be_in.Fobjs)
    for f in be_df.Fobjs:
        f.dev.set_mo_route(0)
    assert abs(be_df.oneshot()[0] - e_df) < 1e-10
    assert all(f.dev.mo_route_used() == (False, j2c.shape[0]) for f in be_df.Fobjs)
    # the same integrals handed over as the reference's semi-sparse tensor (every AO pair stored, no MO screening: eps = 0)
    from quemb_amd import eri_transform as et
    t = et.SemiSparseSym3DTensor.from_dense(np.ascontiguousarray(pqL.transpose(2, 0, 1)), [list(range(N))] * N)
    be_sp = BE(mf, fobj, int_transform="sparse-DF-hip", distribute=False,
               df_ints=dict(int_P_mu_nu=t, j2c=j2c, S_abs=np.abs(mf.get_ovlp()), MO_coeff_epsilon=0.0))
    assert abs(be_sp.oneshot()[0] - e_in) < 1e-6
    assert all(f.dev.mo_route_used() == (True, j2c.shape[0]) for f in be_sp.Fobjs)      # the semi-sparse transform hands its factor over as well


def _semisparse_case(seed=9):
    rng = np.random.default_rng(seed)
    N, n, naux = 12, 5, 40
    # a banded "overlap" so that both screenings bite: AO pairs |mu-nu| > 5 are not stored
    stored = np.abs(np.subtract.outer(np.arange(N), np.arange(N))) <= 5
    L = rng.standard_normal((naux, N, N)); L = (L + L.transpose(0, 2, 1)) * stored
    S_abs = np.exp(-0.9 * np.abs(np.subtract.outer(np.arange(N), np.arange(N))))
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n] * np.exp(-0.8 * np.abs(np.subtract.outer(np.arange(N), 2.0 * np.arange(n))))
    A = rng.standard_normal((naux, naux)); j2c = A @ A.T + naux * np.eye(naux)
    il = np.tril_indices(N)
    return N, n, naux, stored, np.ascontiguousarray(L[:, il[0], il[1]]), S_abs, TA, np.linalg.cholesky(j2c)


def test_semisparse_df_screening_matches_reference_algorithm(qlib):
    from quemb_amd import eri_transform as et
    N, n, naux, stored, packed, S_abs, TA, Lpq = _semisparse_case()
    for eps in (0.0, 1e-3, 5e-2, 0.3):
        ref = oeri.transform_integral_semisparse(packed, stored, TA, S_abs, Lpq, eps)
        df = et.DFContext(L_PQ=Lpq, lib=None)
        df.set_ints(packed, N, "packed")
        got = df.transform(TA, S_abs=S_abs, MO_coeff_epsilon=eps)
        assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), eps
    # the screening really changes the result at a coarse threshold and vanishes at eps = 0
    dense = df.transform(TA)
    assert np.abs(oeri.transform_integral_semisparse(packed, stored, TA, S_abs, Lpq, 0.0) - dense).max() < 1e-11
    assert np.abs(oeri.transform_integral_semisparse(packed, stored, TA, S_abs, Lpq, 0.3) - dense).max() > 1e-6


def test_h8_relaxed_density_sweep_and_octane_relaxed_matching(qlib):
    """relax_density=True through be_func on the device: H8 BE2 sweep against the oracle (Lambda + response densities), then the
    octane BE2 density matching on relaxed densities converges and conserves the electron count of every fragment."""
    from qemb_oracle import be as obe
    from quemb_amd.solver import be_func
    mf, be = _be("h8")
    F = []
    for I, f in enumerate(be.Fobjs):
        o = obe.Frag(f.AO_in_frag, I, f.AO_per_edge, f.ref_frag_idx_per_edge, f.relAO_per_edge, f.relAO_in_ref_per_edge,
                     f.weight_and_relAO_per_center, f.relAO_per_origin)
        obe.init_fragment(o, be.W, be.lmo_coeff, be.Nocc, be.hcore, be.S, be.C, be.hf_dm, be.hf_veff, mf._eri)
        F.append(o)
    err, vec, (ecorr, comps) = be_func(None, be.Fobjs, be.Nocc, "CCSD", be.enuc, eeval=True, return_vec=True, relax_density=True, opts=be.opts)
    err_o, vec_o, (ecorr_o, comps_o) = obe.be_func(None, F, be.Nocc, eeval=True, return_vec=True, relax_density=True)
    assert abs(ecorr - ecorr_o) < 1e-8 and np.allclose(comps, comps_o, atol=1e-8)
    assert np.abs(np.asarray(vec) - np.asarray(vec_o)).max() < 1e-7
    mf8, be8 = _be("octane")
    # (with the reference's HF Jacobian the relaxed problem needs ~100 quasi-Newton iterations for 1e-6; stop at 2e-5 here)
    opt = be8.optimize(solver="CCSD", only_chem=False, relax_density=True, conv_tol=2e-5)
    assert opt.err < 2e-5
    for f in be8.Fobjs:
        assert abs(np.trace(f.rdm1__) - 2 * f.nsocc) < 1e-8
    # relaxed and unrelaxed matched energies differ in the 4th decimal for octane; both are near the golden of the latter
    assert abs(be8.e_corr - (-0.5499514850769742)) < 5e-3


def test_semisparse_tensor_storage_is_consumed_as_is(qlib):
    """Row a5 on the reference's own storage: a SemiSparseSym3DTensor (unique aux vectors + exch_reachable_with_offsets,
    _cpp/eri_sparse_DF.cpp:110-298) goes to the device unexpanded and `transform_integral` (:739-751) runs on it -- the irregular
    first contraction as gathered, batched GEMMs.  Checked against the literal loops of the reference on the same storage, against
    the dense-with-zeros path, and for the layout conventions of the mirror class (offset order, unique = nu <= mu)."""
    from quemb_amd import eri_transform as et
    N, n, naux, stored, packed, S_abs, TA, Lpq = _semisparse_case()
    il = np.tril_indices(N)
    full = np.zeros((naux, N, N)); full[:, il[0], il[1]] = packed; full[:, il[1], il[0]] = packed
    reach = [[int(nu) for nu in np.nonzero(stored[mu])[0]] for mu in range(N)]
    t = et.SemiSparseSym3DTensor.from_dense(full, reach)
    assert t.unique_dense_data.shape == (naux, sum(len(r) for r in t.exch_reachable_unique)) and t.unique_dense_data.flags.f_contiguous
    assert t.offsets[et.ravel_symmetric(0, 0)] == 0 and t.offsets[et.ravel_symmetric(1, 0)] == 1 and t.offsets[et.ravel_symmetric(1, 1)] == 2
    assert all(nu <= mu for mu, r in enumerate(t.exch_reachable_unique) for nu in r)
    assert np.array_equal(t.get_aux_vector(3, 7), full[:, 7, 3])
    df = et.DFContext(L_PQ=Lpq, lib=None)
    df.set_ints_semisparse(t)
    dfd = et.DFContext(L_PQ=Lpq, lib=None)
    dfd.set_ints(packed, N, "packed")
    for eps in (0.0, 1e-3, 5e-2, 0.3):
        ref = oeri.transform_integral_semisparse_csr(t.unique_dense_data, t.exch_reachable_with_offsets, TA, S_abs, Lpq, eps)
        got = df.transform(TA, S_abs=S_abs, MO_coeff_epsilon=eps)
        tol = 1e-11 * max(1.0, np.abs(ref).max())
        assert np.abs(got - ref).max() < tol, eps
        assert np.abs(got - dfd.transform(TA, S_abs=S_abs, MO_coeff_epsilon=eps)).max() < tol
        assert np.abs(ref - oeri.transform_integral_semisparse(packed, stored, TA, S_abs, Lpq, eps)).max() < tol
    # no MO screening: the plain DF transform of the stored pairs
    assert np.abs(df.transform(TA) - dfd.transform(TA)).max() < 1e-11 * np.abs(ref).max()
    # a larger, ragged case (lists of very different lengths, several gather blocks are not needed but empty lists are)
    rng = np.random.default_rng(5)
    N2, n2, naux2 = 40, 9, 23
    stored2 = np.abs(np.subtract.outer(np.arange(N2), np.arange(N2))) <= rng.integers(0, 9, N2)[:, None]
    stored2 = stored2 & stored2.T
    stored2[17, :] = stored2[:, 17] = False                               # an AO without any partner
    L2 = rng.standard_normal((naux2, N2, N2)); L2 = (L2 + L2.transpose(0, 2, 1)) * stored2
    t2 = et.SemiSparseSym3DTensor.from_dense(L2, [[int(x) for x in np.nonzero(stored2[mu])[0]] for mu in range(N2)])
    TA2 = np.linalg.qr(rng.standard_normal((N2, N2)))[0][:, :n2]
    S2 = np.exp(-0.5 * np.abs(np.subtract.outer(np.arange(N2), np.arange(N2))))
    A = rng.standard_normal((naux2, naux2)); Lc = np.linalg.cholesky(A @ A.T + naux2 * np.eye(naux2))
    df2 = et.DFContext(L_PQ=Lc, lib=None)
    df2.set_ints_semisparse(t2)
    import os
    for eps, budget in ((0.0, None), (0.2, None), (0.2, "700"), (0.0, "1")):     # small budgets: several gather blocks, down to one AO each
        ref2 = oeri.transform_integral_semisparse_csr(t2.unique_dense_data, t2.exch_reachable_with_offsets, TA2, S2, Lc, eps)
        if budget:
            os.environ["QEMB_DF_GATHER_BUDGET"] = budget
        try:
            got2 = df2.transform(TA2, S_abs=S2, MO_coeff_epsilon=eps)
        finally:
            os.environ.pop("QEMB_DF_GATHER_BUDGET", None)
        assert np.abs(got2 - ref2).max() < 1e-11 * max(1.0, np.abs(ref2).max()), (eps, budget)
    # embedding orbitals localised on a stretch of the AOs: distant AOs are reached by no orbital and leave both contractions
    TA3 = np.zeros((N2, n2)); TA3[5:20] = np.linalg.qr(rng.standard_normal((15, n2)))[0]
    n_act = int((np.abs(S2 @ TA3) >= 0.05).any(axis=1).sum())
    assert 0 < n_act < N2
    ref3 = oeri.transform_integral_semisparse_csr(t2.unique_dense_data, t2.exch_reachable_with_offsets, TA3, S2, Lc, 0.05)
    assert np.abs(df2.transform(TA3, S_abs=S2, MO_coeff_epsilon=0.05) - ref3).max() < 1e-11 * max(1.0, np.abs(ref3).max())
    assert np.abs(df2.transform(TA3, S_abs=S2, MO_coeff_epsilon=1e9)).max() == 0.0          # everything screened away
    with pytest.raises(ValueError):
        et.SemiSparseSym3DTensor((naux, N, N), [[1], []] + [[] for _ in range(N - 2)])       # not symmetric
    with pytest.raises(ValueError):
        df.set_ints_semisparse(et.SemiSparseSym3DTensor((naux, N, N), reach))                  # unfilled (NaN) data


def test_octane_chemical_potential_goldens_be2_be3(qlib):
    """tests/chempot_molBE_test.py:51-65: octane/STO-3G, autogen BE2 and BE3, chemical-potential-only matching
    (`only_chem=True`): E_tot = -310.33471581 / -310.33447096, delta 1e-4."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    mol = Mole(GOLDEN / "octane.xyz")
    mf = RHF(mol); mf.kernel()
    for key, target in (("test_autogen_octane_be2", -310.33471581), ("test_autogen_octane_be3", -310.33447096)):
        be = BE(mf, FragPart.from_json(GOLDEN / "fragmentation.json", key), distribute=False)
        assert abs(be.ebe_hf - mf.e_tot) < 1e-6                 # HF-in-HF (tests/hf-in-hf_BE_test.py:56-63, 1e-5)
        be.optimize(solver="CCSD", only_chem=True)
        assert abs(be.ebe_tot - target) < 1e-4, (key, be.ebe_tot, target)


def test_octane_be3_density_matching_golden(qlib):
    """tests/molbe_octane_test.py:43-68 (the reference's QUEMB_DO_EXPENSIVE_TESTS case): BE3 CCSD density matching of octane,
    E_tot = -310.3344717358742, E_corr = -0.5497021857717073 (np.isclose, rtol 1e-5; PySCF conv_tol 1e-7)."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    mf = RHF(Mole(GOLDEN / "octane.xyz")); mf.kernel()
    be = BE(mf, FragPart.from_json(GOLDEN / "fragmentation.json", "test_autogen_octane_be3"), distribute=False)
    opt = be.optimize(solver="CCSD", only_chem=False)
    assert opt.err < 1e-6
    assert abs(be.e_corr - (-0.5497021857717073)) < 2e-6
    assert abs(be.ebe_tot - (-310.3344717358742)) < 2e-6


def test_octane_numerical_jacobian_matching(qlib):
    """tests/numerical_jac_test.py:24-44 in spirit (octane, BE2, `jac_solver="Numerical"` against `"HF"`, atol 1e-5): the
    281 x 281 central-difference CCSD Jacobian (2 x 280 single-fragment solves + 2 sweeps) and the HF (CPHF) Jacobian lead the
    QN optimisation to the same energy."""
    import time
    mf, be = _be("octane")
    t0 = time.time()
    be.optimize(solver="CCSD", jac_solver="Numerical", step_size=1e-4)
    t_num = time.time() - t0
    mf2, be2 = _be("octane")
    t0 = time.time()
    be2.optimize(solver="CCSD", jac_solver="HF")
    print(f"octane BE2 matching: numerical Jacobian {t_num:.2f} s / {be.beopt.iter} QN iterations, HF Jacobian {time.time() - t0:.2f} s / {be2.beopt.iter}")
    assert be.beopt.err < 1e-6 and abs(be.ebe_tot - be2.ebe_tot) < 2e-6
    assert abs(be.ebe_tot - (-310.3347211309688)) < 5e-6         # tests/molbe_octane_test.py:32-36


def test_hf_in_hf_reference_cases(qlib):
    """tests/hf-in-hf_BE_test.py:16-63, every molecular case: H8/STO-3G, H8/cc-pVDZ and octane/STO-3G, autogen BE1, BE2, BE3:
    `ebe_hf == mf.e_tot` (reference delta 1e-5; here 1e-7)."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    h8 = [["H", (0.0, 0.0, float(i))] for i in range(8)]
    for mol, stem, rep in ((Mole(h8), "test_autogen_h_linear_be", 1), (Mole(h8, basis="cc-pvdz"), "test_autogen_h_linear_be", 5),
                           (Mole(GOLDEN / "octane.xyz"), "test_autogen_octane_be", 1)):
        mf = RHF(mol); mf.kernel()
        for n_BE in (1, 2, 3):
            fobj = FragPart.from_json(GOLDEN / "fragmentation.json", f"{stem}{n_BE}", n_BE=n_BE)
            be = BE(mf, fobj.replicate_sites(rep) if rep > 1 else fobj, distribute=False)
            assert abs(be.ebe_hf - mf.e_tot) < 1e-7, (stem, rep, n_BE, be.ebe_hf - mf.e_tot)


def test_octane_frozen_core_density_matching_golden(qlib):
    """tests/molbe_octane_get_rdms_test.py:52-67: octane/STO-3G BE2 CCSD density matching with the frozen-core approximation,
    chemgen and autogen fragmentations alike: E_tot = -310.3311676424482 (np.isclose, rtol 1e-5).  The chemgen lists are the
    reference's own expected data (tests/golden/fragmentation_chemgen.json); the autogen ones are the all-electron fixture
    with the core AOs dropped (`FragPart.freeze_core`, the rule of molbe/autofrag.py:519-548)."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    mol = Mole(GOLDEN / "octane.xyz")
    mf = RHF(mol); mf.kernel()
    for fobj in (FragPart.from_json(GOLDEN / "fragmentation_chemgen.json", "chemgen_octane_be2_frozen_core"),
                 FragPart.from_json(GOLDEN / "fragmentation.json", "test_autogen_octane_be2").freeze_core(mol)):
        assert fobj.frozen_core
        be = BE(mf, fobj, distribute=False)
        assert be.ncore == 8 and be.W.shape == (58, 50) and be.Nocc == 25
        assert abs(be.ebe_hf - mf.e_tot) < 1e-7                 # HF-in-HF with E_core (mbe.py:1170)
        opt = be.optimize(solver="CCSD", only_chem=False)
        assert opt.err < 1e-6
        assert abs(be.ebe_tot - (-310.3311676424482)) < 5e-6, be.ebe_tot
        D_ao = be.rdm1_fullbasis(return_ao=True, only_rdm1=True)     # :59 of the reference test; the 50 valence electrons
        assert abs(np.trace(D_ao @ be.S) - 50.0) < 1e-4


def test_periodic_front_end_matches_reference_goldens(qlib):
    """kbe/pfrag.py:143-306 (k -> R Fourier, SVD Schmidt, cons_h1, get_nsocc) on the device against the reference's own outputs."""
    from helpers import check_periodic_front_end
    check_periodic_front_end(None)


def test_periodic_fragment_sweep(qlib):
    """kbe/pfrag.py:240-268 (cons_fock through kbe/helper.py get_veff) and the inherited scf / update_ebe_hf / sweep body on a
    periodic fragment: k sums against a NumPy restatement, the solve against the oracle."""
    from helpers import check_periodic_fragment_sweep
    check_periodic_fragment_sweep(None)


def test_concurrent_streams_give_identical_results(qlib):
    """nstreams > 1: fragments driven from several host threads, each bound to its own execution context (HIP stream,
    workspaces, block cache).  Results must be bit-for-bit those of the serial sweep (every kernel is deterministic and
    nothing is shared between contexts)."""
    from quemb_amd.solver import be_func
    mf, be1 = _be("octane")
    mf3, be3 = _be("octane", nstreams=3)
    r1 = be_func(None, be1.Fobjs, be1.Nocc, "CCSD", be1.enuc, eeval=True, return_vec=True, opts=be1.opts)
    r3 = be_func(None, be3.Fobjs, be3.Nocc, "CCSD", be3.enuc, eeval=True, return_vec=True, opts=be3.opts, nstreams=3)
    assert r1[0] == r3[0] and np.array_equal(np.asarray(r1[1]), np.asarray(r3[1]))
    assert r1[2][0] == r3[2][0]
    for a, b in zip(be1.Fobjs, be3.Fobjs):
        assert np.array_equal(a._rdm1, b._rdm1) and np.array_equal(a.t1, b.t1)
    # a second pass (warm block caches, contexts reused) and the optimiser on top of it.  (The fragment RHF starts its Jacobi
    # eigensolver in the orbitals of the fragment's previous solve, so a REPEATED sweep agrees to rounding, not bit for bit;
    # serial against concurrent, above, is bitwise.)
    r3b = be_func(None, be3.Fobjs, be3.Nocc, "CCSD", be3.enuc, eeval=True, return_vec=True, opts=be3.opts, nstreams=3)
    r1b = be_func(None, be1.Fobjs, be1.Nocc, "CCSD", be1.enuc, eeval=True, return_vec=True, opts=be1.opts)
    assert abs(r3b[2][0] - r3[2][0]) < 1e-13 and r3b[2][0] == r1b[2][0]
    opt = be3.optimize(solver="CCSD", only_chem=False)
    assert opt.err < 1e-6 and abs(be3.e_corr - (-0.5499514850769742)) < 5e-6


def test_h8_ccpvdz_hf_in_hf_and_oneshot_be2_vs_molecular_ccsd(qlib):
    """H8 / cc-pVDZ (N = 40, p functions, 5 AOs per site): HF-in-HF (tests/hf-in-hf_BE_test.py:56-63) and the one-shot BE2 CCSD
    correlation energy against the CCSD of the whole molecule run through the same device solver (BE2 is within 1e-3 Eh)."""
    from quemb_amd.fragpart import FragPart
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    mol = Mole([["H", (0.0, 0.0, float(i))] for i in range(8)], basis="cc-pvdz")
    mf = RHF(mol); mf.kernel()
    N = mol.nao
    w, U = np.linalg.eigh(mf.get_ovlp()); W = U @ np.diag(w ** -0.5) @ U.T
    fr = DeviceFragment(N, N)
    et.AOEri(mf._eri, N).transform(W, frag=fr, want_host=False)
    out = fr.solve(mol.nelectron // 2, W.T @ mf.get_hcore() @ W, opts=default_opts(), eeval=False)
    assert abs(out["e_scf"] + mf.energy_nuc() - mf.e_tot) < 1e-9            # the device RHF reproduces the molecular RHF
    be = BE(mf, FragPart.from_json(GOLDEN / "fragmentation.json", "test_autogen_h_linear_be2").replicate_sites(5), distribute=False)
    assert abs(be.hf_err) < 1e-9
    e, _ = be.oneshot()
    assert abs(e - out["e_corr_mo"]) < 2e-3, (e, out["e_corr_mo"])


def test_mp2_and_ccsd_model_jacobians_match_reference(qlib):
    """jac_solver="MP2" / "CCSD" (optqn.py:437-461; jac_utils.py:162-178, cpmp2_utils.py:94-133): the density responses assembled
    from device-exported MO blocks == the reference's own functions (tests/golden/jac.npz)."""
    from test_hostlogic_be import _check_jacobian_models
    _check_jacobian_models(qlib)


def test_octane_matching_with_correlated_model_jacobians(qlib):
    """Octane BE2 density matching started from the MP2- and CCSD-model Jacobians reaches the reference's golden energy
    (tests/molbe_octane_test.py:32-36) like the HF Jacobian; how far each initial Jacobian is from the exact (central-difference)
    CCSD one is printed."""
    from quemb_amd.jacobian import get_be_error_jacobian
    mf, be = _be("octane")
    Jn = be.compute_numerical_jacobian("CCSD", False, 1, step_size=1e-4)
    for js in ("HF", "MP2", "CCSD"):
        J0 = get_be_error_jacobian(be.fobj.n_frag, be.Fobjs, jac_solver=js, opts=be.opts)
        print(f"octane BE2 initial Jacobian {js:4s}: |J0 - J_numerical|_max = {np.abs(J0 - Jn).max():.3e}  (|J_numerical|_max = {np.abs(Jn).max():.3e})")
    for js in ("MP2", "CCSD"):
        mf2, be2 = _be("octane")
        be2.optimize(solver="CCSD", jac_solver=js)
        print(f"octane BE2 matching with the {js} Jacobian: {be2.beopt.iter} QN iterations")
        assert be2.beopt.err < 1e-6
        assert abs(be2.ebe_tot - (-310.3347211309688)) < 5e-6


def test_periodic_direct_df_matches_reference_outputs(qlib):
    """kbe/eri_onthefly.py:19-45, :48-241 (Gamma-point CC-GDF: Cholesky / eigenvalue fit of the metric, plane-wave and real-space
    accumulation, fit, bb^T bb, imaginary-part test) on the device against the reference's own outputs (tests/golden/kbe_df.npz)."""
    from test_kbe_df import check_periodic_df, check_periodic_df_argument_errors
    check_periodic_df(qlib)
    check_periodic_df_argument_errors(qlib)


def test_periodic_direct_df_beyond_one_tile_and_into_a_fragment(qlib):
    """the same path at sizes past one GEMM tile (indefinite metric: eigenvalue fit through the device Jacobi solver), against the
    oracle restatement; the ERIs written straight into a device fragment are the ones returned to the host."""
    from qemb_oracle import kbe_df
    from quemb_amd import kbe_eri_onthefly as keo
    from quemb_amd.fragsolver import DeviceFragment
    from test_kbe_df import big_case
    src, TAs = big_case()
    ref, ischol = kbe_df.integral_direct_DF(src, TAs, 16, 7)
    assert not ischol

    class F:
        pass
    frs = []
    for TA in TAs:
        f = F(); f.TA = TA; f.dev = DeviceFragment(TA.shape[1], 3, lib=qlib); frs.append(f)
    host = keo.integral_direct_DF(src, frs, pw_step=32, aux_step=25, lib=qlib, want_host=True)
    assert keo.integral_direct_DF(src, frs, pw_step=1000, aux_step=1000, lib=qlib) is None
    for f, e, r in zip(frs, host, ref):
        assert np.abs(e - r).max() < 1e-9 * np.abs(r).max()
        assert np.abs(e - e.T).max() < 1e-12 * np.abs(r).max()
        assert np.abs(f.dev.get_eri_s4() - r).max() < 1e-9 * np.abs(r).max()
        f.dev.free()


def test_periodic_driver_matches_the_supercell(qlib):
    """BASELINE configs[4] (periodic BE2 with k-points) on the model the image allows: the k-point driver (kbe_pbe.BE, mirror of
    kbe/pbe.py) on a ring of 4 cells x 3 orbitals with 4 k-points against the molecular driver on the 12-orbital supercell -- HF-in-HF,
    one-shot and density-matched CCSD energies per cell, matched potentials; then the Gamma-point int-direct-DF branch."""
    from test_kbe_pbe import check_gamma_point_driver_with_direct_df, check_periodic_driver
    check_periodic_driver(qlib)
    check_periodic_driver(qlib, nk=3, nlo=3)
    check_gamma_point_driver_with_direct_df(qlib)


def test_c5_dimensions_kpoint_view_equals_supercell_view(qlib):
    """BASELINE configs[4] AT ITS OWN DIMENSIONS (kbe polyacetylene BE2: 24 AOs per cell, 28 electrons, 1 x 1 x 3 k-points; the reference's
    libdmet / PySCF-PBC integrals do not exist here, so the integrals are a density-fitted model of that size): kbe_pbe.BE with the four BE2
    fragments of the reference cell (18 sites + 18 bath orbitals), fragment ERIs from the supercell's DF integrals through the device CC-GDF
    transform (kbe_eri_onthefly.integral_direct_DF), against the molecular driver on the 72-orbital supercell with 12 fragments and dense
    integrals -- HF-in-HF 1e-8, fragment ERIs 1e-10, one-shot E_corr per cell and its pieces 1e-8, density-matched E_corr per cell 5e-7 and
    the translational symmetry of the 169 matched potentials.  kbe/pbe.py:78-316, :502-716."""
    from test_kbe_pbe import check_c5_driver
    m, kbe, mol = check_c5_driver(qlib, matching=True)
    assert kbe.nstreams is not None and kbe.lockstep is True      # the sweep mode was chosen from the fragments' sizes (four fragments of 36 orbitals: lock step)


def test_whole_system_fragment_is_the_molecular_ccsd(qlib):
    from test_hostlogic_be import check_schmidt_svd_wide_and_empty_environment, check_whole_system_fragment
    check_whole_system_fragment(qlib)
    check_schmidt_svd_wide_and_empty_environment(qlib)


def test_abs_overlap_quadrature_and_reachability_on_the_device(qlib):
    """approx_S_abs (molbe/eri_sparse_DF.py:928-959) with the primitive quadrature on the device, _get_AO_per_AO (:224-240):
    against the restatement of oracle/qemb_oracle/sparse_df.py (itself checked against a grid integral, tests/test_oracle_sparse_df.py)."""
    from test_oracle_sparse_df import check_abs_overlap_and_reachability
    check_abs_overlap_and_reachability(qlib)


def test_sparse_df_from_geometry_h8_and_octane(qlib):
    """int_transform = "sparse-DF-hip" / "on-fly-sparse-DF-hip" / "int-direct-DF-hip" with an auxiliary basis and nothing else (the
    reference's sparse-DF(-gpu) / on-fly-sparse-DF(-gpu) / int-direct-DF branches, mbe.py:1049-1110): auxiliary molecule, (P|Q), AO
    screening, semi-sparse (P|mu nu) fill, device transform.  H8 and octane / STO-3G with even-tempered auxiliary sets up to d on H and
    f on C (the reference's own `weigend` goldens, tests/test_eri_sparse_DF.py:32-54, need the def2-universal-jfit table, which this
    image does not hold)."""
    from helpers import GOLDEN
    from test_oracle_sparse_df import check_sparse_df_from_geometry
    errs = check_sparse_df_from_geometry(qlib)
    print(f"H8 BE2 one-shot: |E(DF) - E(in-core)| = {errs[0]:.3e} (s aux on H) -> {errs[1]:.3e} (s,p,d aux on H)")
    errs = check_sparse_df_from_geometry(qlib, atoms=str(GOLDEN / "octane.xyz"), frag_key="test_autogen_octane_be2", tol=1e-8)
    print(f"octane BE2 one-shot: |E(DF) - E(in-core)| = {errs[0]:.3e} (H: s, C: s,p) -> {errs[1]:.3e} (H: s,p,d, C: s,p,d,f)")


def test_transforms_at_survey_sizes(qlib):
    """BASELINE configs[3] at its full size: the dense AO -> fragment transform at N_ao = 256 (s8 input) and the
    density-fitted one at N_ao = 512, n_aux = 1000, both to an n = 220 fragment, against a NumPy evaluation of the same integrals
    through their factorised form (ij|kl) = sum_P b_P,ij b_P,kl, b = TA^T B TA (with (P|Q)^-1/2 folded in for DF)."""
    from quemb_amd import eri_transform as et
    rng = np.random.default_rng(20260803)
    n = 220
    iln = np.tril_indices(n)
    # ---- a3: dense, 8-fold packed input
    N = 256
    npair = N * (N + 1) // 2
    B = 0.06 * rng.standard_normal((64, npair))
    s4 = B.T @ B
    s8 = s4[np.tril_indices(npair)]
    del s4
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n].copy()
    ao = et.AOEri(s8, N, lib=qlib)
    out = ao.transform(TA, want_host=True)
    ao.free()
    il = np.tril_indices(N)
    Bf = np.zeros((64, N, N)); Bf[:, il[0], il[1]] = B; Bf = Bf + Bf.transpose(0, 2, 1); Bf[:, np.arange(N), np.arange(N)] *= 0.5
    b = np.einsum("Ppq,pi,qj->Pij", Bf, TA, TA, optimize=True)[:, iln[0], iln[1]]
    ref = b.T @ b
    assert np.abs(out - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    # ---- a4: density fitted, n_aux = 1000
    N, naux = 512, 1000
    npair = N * (N + 1) // 2
    ints = 0.06 * rng.standard_normal((naux, npair))
    A = rng.standard_normal((naux, naux)) / np.sqrt(naux)
    j2c = A @ A.T + np.eye(naux)
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n].copy()
    df = et.DFContext(j2c=j2c, lib=qlib)
    df.set_ints(ints, N, layout="packed")
    out = df.transform(TA, want_host=True)
    df.free()
    il = np.tril_indices(N)
    Bf = np.zeros((naux, N, N)); Bf[:, il[0], il[1]] = ints; Bf = Bf + Bf.transpose(0, 2, 1); Bf[:, np.arange(N), np.arange(N)] *= 0.5
    b = np.einsum("Ppq,pi,qj->Pij", Bf, TA, TA, optimize=True)[:, iln[0], iln[1]]
    bp = np.linalg.solve(np.linalg.cholesky(j2c), b)
    ref = bp.T @ bp
    assert np.abs(out - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_semisparse_storage_at_survey_size(qlib):
    """Row a5 at the size SURVEY 8(d) names for it (N_ao = 512, n_aux = 1000, fragment n = 220) on the SEMI-SPARSE storage itself: a banded
    SemiSparseSym3DTensor (unique aux vectors + exch_reachable_with_offsets, ~25 partners per AO, two AOs without any) goes to the device
    unexpanded; the transform -- plain, and with the reference's MO screening -- equals the one of the same integrals stored dense with zeros."""
    from quemb_amd import eri_transform as et
    rng = np.random.default_rng(20261004)
    N, n, naux = 512, 220, 1000
    width = rng.integers(4, 48, N)
    d = np.abs(np.subtract.outer(np.arange(N), np.arange(N)))
    stored = (d <= width[:, None]) & (d <= width[None, :])
    stored[[100, 333], :] = False; stored[:, [100, 333]] = False
    reach = [[int(nu) for nu in np.nonzero(stored[mu])[0]] for mu in range(N)]
    il = np.tril_indices(N)
    keep = stored[il]
    packed = np.zeros((naux, il[0].size))
    packed[:, keep] = 0.06 * rng.standard_normal((naux, int(keep.sum())))
    # the mirror class from its sparse parts (no (naux, N, N) array is ever formed): unique pairs nu <= mu in offset order
    t = et.SemiSparseSym3DTensor((naux, N, N), reach)
    pair = il[0] * (il[0] + 1) // 2 + il[1]                       # == ravel_symmetric(mu, nu), nu <= mu
    cols = np.array([t.offsets[int(p)] for p in pair[keep]])
    assert cols.min() == 0 and np.unique(cols).size == cols.size == t.unique_dense_data.shape[1]
    t.unique_dense_data[:, cols] = packed[:, keep]
    A = rng.standard_normal((naux, naux)) / np.sqrt(naux)
    j2c = A @ A.T + np.eye(naux)
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n].copy()
    S_abs = np.exp(-d / 6.0)
    df = et.DFContext(j2c=j2c, lib=qlib); df.set_ints_semisparse(t)
    dfd = et.DFContext(j2c=j2c, lib=qlib); dfd.set_ints(packed, N, layout="packed")
    a, b = df.transform(TA, want_host=True), dfd.transform(TA, want_host=True)
    scale = max(1.0, np.abs(b).max())
    assert np.abs(a - b).max() < 1e-11 * scale
    for eps in (1e-5, 3e-2):
        a, b2 = df.transform(TA, S_abs=S_abs, MO_coeff_epsilon=eps), dfd.transform(TA, S_abs=S_abs, MO_coeff_epsilon=eps)
        assert np.abs(a - b2).max() < 1e-11 * scale
    assert np.abs(b2 - b).max() > 1e-9 * scale          # the coarser threshold did drop contributions
    df.free(); dfd.free()


def test_rccl_all_reduce_branch_single_rank(qlib):
    """The `nccl` (= RCCL) branch of be_parallel.all_reduce_sum -- device tensor, all-reduce, failure slot -- on a one-rank process
    group (the boxes of this pool have one GPU; the 8-GPU run is the driver's): values survive the round trip and a rank-local
    failure comes back as RankFailure."""
    import socket
    import torch
    import torch.distributed as dist
    from quemb_amd import be_parallel
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        buf = np.arange(37, dtype=np.float64) * 0.5
        keep = buf.copy()
        be_parallel.all_reduce_sum(buf, force=True)
        assert np.array_equal(buf, keep)
        with pytest.raises(be_parallel.RankFailure):
            be_parallel.all_reduce_sum(buf, error=ValueError("fragment 3 did not converge"), force=True)
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()


def test_library_rccl_communicator_single_rank(qlib):
    """qemb_comm_* on the real RCCL (include/qemb_hip.h): id, communicator on the library's device and stream, sum / max all-reduce of
    host buffers through the pinned staging path (also one longer than the first staging allocation), be_parallel picking the
    communicator up, the failure slot, destroy.  One rank: the boxes of this pool have one GPU; N > 1 is the driver's 8-GPU run."""
    from quemb_amd import be_parallel, comm
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert comm.info(qlib) == (0, 1) and comm.active() is None
    uid = comm.unique_id(qlib)
    assert len(uid) == comm.ID_BYTES and uid != bytes(comm.ID_BYTES)
    comm.init(qlib, 0, 1, uid)
    try:
        assert comm.info(qlib) == (0, 1)
        x = np.linspace(-3.0, 7.0, 341)
        keep = x.copy()
        assert np.array_equal(comm.all_reduce(qlib, x), keep)
        assert np.array_equal(comm.all_reduce(qlib, x, comm.MAX), keep)
        big = np.random.default_rng(3).standard_normal(200_000)
        assert np.array_equal(comm.all_reduce(qlib, big.copy()), big)
        comm.barrier(qlib)
        with pytest.raises(_lib.QembError):
            comm.init(qlib, 0, 1, uid)                       # one communicator per process
        comm._force_single = True                            # run be_parallel's collective on the one-rank communicator
        try:
            assert be_parallel.world() == (0, 1)
            buf = np.arange(37, dtype=np.float64) * 0.5
            be_parallel.all_reduce_sum(buf)
            assert np.array_equal(buf, np.arange(37) * 0.5)
            with pytest.raises(be_parallel.RankFailure):
                be_parallel.all_reduce_sum(buf, error=ValueError("fragment 3 did not converge"))
        finally:
            comm._force_single = False
    finally:
        comm.destroy(qlib)
    assert comm.active() is None
    with pytest.raises(_lib.QembError):
        comm.all_reduce(qlib, np.zeros(3))                   # no communicator: loud


_MISSING_PEER_SCRIPT = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, sys.argv[1])
from quemb_amd import _lib, comm
lib = _lib.init(0)
uid = comm.unique_id(lib)
t0 = time.monotonic()
try:
    comm.init(lib, 0, 2, uid)            # world of two, the second rank never starts
    print("JOINED", flush=True)
except _lib.QembError as e:
    print(f"TIMEOUT {time.monotonic() - t0:.1f} {e}", flush=True)
try:
    comm.all_reduce(lib, __import__("numpy").zeros(2))
    print("REDUCED", flush=True)
except _lib.QembError as e:
    print(f"LOUD {e}", flush=True)
sys.stdout.flush()
os._exit(0)                               # the helper thread is still inside ncclCommInitRank: leave without running exit handlers
"""


@pytest.mark.timeout(180)
def test_library_rccl_rendezvous_is_bounded(qlib):
    """A peer that never joins is an error after QEMB_COMM_TIMEOUT_S, not a hang (csrc/comm_rccl.hip, round 4): rank 0 of a world of
    two on the real RCCL, in a child process (its rendezvous thread never returns)."""
    import subprocess
    import sys
    env = dict(os.environ, QEMB_COMM_TIMEOUT_S="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _MISSING_PEER_SCRIPT, str(ROOT)], env=env, capture_output=True, text=True, timeout=150)
    out = p.stdout
    assert "JOINED" not in out and "REDUCED" not in out, out + p.stderr[-1500:]
    line = [ln for ln in out.splitlines() if ln.startswith("TIMEOUT")]
    assert line, out + p.stderr[-1500:]
    waited = float(line[0].split()[1])
    assert 7.0 <= waited <= 30.0 and "QEMB_COMM_TIMEOUT_S" in line[0]
    assert any(ln.startswith("LOUD") for ln in out.splitlines())


def test_df_transform_matches_reference_integral_direct_DF(qlib):
    """row a4 on the HIP library against the outputs of the reference's own integral_direct_DF (tests/golden/df.npz)"""
    from test_df_golden import check_df_golden
    check_df_golden(qlib)


def test_lockstep_tapes_are_kept_between_sweeps_and_equal_fresh_recordings(qlib, monkeypatch):
    """A fragment keeps the recorded amplitude update of a lock-step sweep for its next solve when that solve's buffers land where the last one's did
    (same allocation trace).  With QEMB_TAPE_CACHE_CHECK=1 every kept tape is compared, launch by launch and argument byte by argument byte, with a fresh
    recording; the sweeps give the numbers of the first one; a sweep at a different potential (other numbers in the same buffers) those of a serial sweep of a fresh BE object."""
    import ctypes as C
    from quemb_amd.solver import be_func
    monkeypatch.setenv("QEMB_TAPE_CACHE_CHECK", "1")
    mfl, bel = _be("octane", lockstep=True)
    reused, recorded = C.c_int64(), C.c_int64()
    qlib.qemb_tape_cache_counters(C.byref(reused), C.byref(recorded), 1)
    first = None
    for sweep in range(4):
        r = be_func(None, bel.Fobjs, bel.Nocc, "CCSD", bel.enuc, eeval=True, return_vec=True, opts=bel.opts, lockstep=True)
        if first is None:
            first = (r[0], np.asarray(r[1]).copy(), [f._rdm1.copy() for f in bel.Fobjs])
        else:
            # (not bit for bit: from the second sweep on the fragment RHF starts in the orbitals of the sweep before)
            assert abs(r[0] - first[0]) < 1e-12 and np.abs(np.asarray(r[1]) - first[1]).max() < 1e-11
            assert all(np.abs(f._rdm1 - d).max() < 1e-10 for f, d in zip(bel.Fobjs, first[2]))
    qlib.qemb_tape_cache_counters(C.byref(reused), C.byref(recorded), 0)
    nfr = len(bel.Fobjs)
    assert recorded.value >= nfr and reused.value >= nfr, (reused.value, recorded.value)      # the buffer layout settles after a sweep or two; from then on no recording
    assert reused.value + recorded.value == 4 * nfr
    # other numbers in the same buffers: a kept tape must serve a sweep at another potential exactly like a fresh recording
    monkeypatch.setenv("QEMB_TAPE_CACHE_CHECK", "0")
    pot = np.array([0.01 * (k % 3 - 1) for k in range(len(first[1]))])
    ra = be_func(pot, bel.Fobjs, bel.Nocc, "CCSD", bel.enuc, eeval=True, return_vec=True, opts=bel.opts, lockstep=True)
    mf1, be1 = _be("octane")
    rb = be_func(pot, be1.Fobjs, be1.Nocc, "CCSD", be1.enuc, eeval=True, return_vec=True, opts=be1.opts)
    assert abs(ra[0] - rb[0]) < 1e-12 and np.abs(np.asarray(ra[1]) - np.asarray(rb[1])).max() < 1e-11
    assert np.abs(np.asarray(ra[1]) - first[1]).max() > 1e-4      # (and it IS another sweep)


def test_lockstep_sweep_gives_identical_results(qlib):
    """lockstep=True: every fragment of the octane BE2 sweep in ONE library call (qemb_frag_solve_batch) -- fragment phases per stream,
    CCSD iterations of all six fragments in lock step with one grouped launch per operation.  Bit for bit the serial sweep; the density
    matching on top of it reproduces the reference's golden energy."""
    from quemb_amd.solver import be_func
    mf, be1 = _be("octane")
    mfl, bel = _be("octane", lockstep=True)
    stats = {}
    r1 = be_func(None, be1.Fobjs, be1.Nocc, "CCSD", be1.enuc, eeval=True, return_vec=True, opts=be1.opts)
    rl = be_func(None, bel.Fobjs, bel.Nocc, "CCSD", bel.enuc, eeval=True, return_vec=True, opts=bel.opts, lockstep=True, stats=stats)
    assert r1[0] == rl[0] and np.array_equal(np.asarray(r1[1]), np.asarray(rl[1]))
    assert r1[2][0] == rl[2][0] and list(r1[2][1]) == list(rl[2][1])
    for a, b in zip(be1.Fobjs, bel.Fobjs):
        assert np.array_equal(a._rdm1, b._rdm1) and np.array_equal(a.t1, b.t1) and np.array_equal(a.mo_coeffs, b.mo_coeffs)
    assert stats["max_group"] == len(bel.Fobjs) and stats["grouped_launches"] > 0
    assert stats["launches"] * 3 < stats["operations"]          # the launch count of the iterations fell by more than 3x
    opt = bel.optimize(solver="CCSD", only_chem=False)
    assert opt.err < 1e-6 and abs(bel.e_corr - (-0.5499514850769742)) < 5e-6
