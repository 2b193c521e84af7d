"""CPU: HOST LOGIC of the drivers (GEMM factorisation of RCCSD, MO transform, SCF loop, energy contraction)
against the oracle, by linking the product's driver sources to the scalar mock device layer
(tests/hostcheck).  The HIP kernels themselves are covered by the -m gpu tests."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import synthetic_fragment
from qemb_oracle import be, ccsd, eri, rdm, scf

sys.path.insert(0, str(Path(__file__).resolve().parent / "hostcheck"))


@pytest.fixture(scope="module")
def hlib():
    import build as hc_build
    from quemb_amd import _lib
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    assert lib.qemb_backend() == b"hostcheck"
    return lib


def _problem(n, o, nf, seed):
    h, e1 = synthetic_fragment(n, o, seed)
    rng = np.random.default_rng(seed + 1)
    h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
    veff0 = rng.standard_normal((n, n)); veff0 = veff0 + veff0.T
    veff = rng.standard_normal((n, n)); veff = veff + veff.T
    return h, e1, h1, veff0, veff


@pytest.mark.parametrize("n,o,nf,cen", [(6, 2, 3, [0, 1]), (8, 3, 3, [1]), (7, 4, 2, [0])])
def test_fragment_solve_matches_oracle(hlib, n, o, nf, cen):
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    h, e1, h1, veff0, veff = _problem(n, o, nf, 40 + n)
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf, lib=hlib)
    fr.set_eri_s4(s4)
    fr.set_energy_data(h1, veff0, veff, 0.75, cen)
    opts = default_opts(hlib, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9)
    out = fr.solve(o, h, opts=opts, eeval=True, want_t2=True)
    # oracle
    mf = scf.rhf(h, e1, o, conv_tol=1e-13, conv_tol_grad=1e-9)
    t1, t2, ecc, nit = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"], conv_tol=1e-13, conv_tol_normt=1e-11)
    assert abs(out["e_scf"] - mf["e_tot"]) < 1e-10
    assert np.abs(out["mo_energy"] - mf["mo_energy"]).max() < 1e-8
    assert abs(out["e_corr_mo"] - ecc) < 1e-10, (out["e_corr_mo"], ecc)
    r1 = rdm.make_rdm1_ccsd_t1(t1)
    rdm_emb = mf["mo_coeff"] @ r1 @ mf["mo_coeff"].T * 0.5
    assert np.abs(out["rdm1_emb"] - rdm_emb).max() < 1e-8
    r2 = rdm.make_rdm2_urlx(t1, t2, with_dm1=False)
    TA = np.zeros((n + 2, n))
    e_ref = be.get_frag_energy(mf["mo_coeff"], o, nf, (0.75, cen), TA, h1, r1, r2, s4, veff0, None, True)
    assert np.allclose(out["e_frag"], e_ref, atol=1e-9), (out["e_frag"], e_ref)
    # HF fragment energy (update_ebe_hf) with the same orbitals
    f = be.Frag(list(range(nf)), 0, [], [], [], [], (0.75, cen))
    f.h1, f.veff, f.TA, f._mo_coeffs, f.nsocc, f.eri_s4 = h1, veff, TA, mf["mo_coeff"], o, s4
    assert abs(out["ebe_hf"] - be.update_ebe_hf(f)) < 1e-9
    # amplitudes up to the MO phase convention: compare through phase-invariant contractions
    C_g, C_o = out["mo_coeff"], mf["mo_coeff"]
    t1_emb_g = C_g[:, :o] @ out["t1"] @ C_g[:, o:].T
    t1_emb_o = C_o[:, :o] @ t1 @ C_o[:, o:].T
    assert np.abs(t1_emb_g - t1_emb_o).max() < 1e-8
    # J/K entry point
    P = np.random.default_rng(1).standard_normal((n, n)); P = P + P.T
    J, K = fr.jk(P)
    Jr, Kr = scf.get_jk(e1, P)
    assert np.abs(J - Jr).max() < 1e-11 and np.abs(K - Kr).max() < 1e-11


def check_factor_route_equals_four_index(lib, cases=((8, 3, 3, 20), (11, 4, 4, 0), (9, 2, 3, 31), (7, 7, 2, 12), (6, 2, 2, 60)), tol=2e-10):
    """MO integrals from the fragment's 3-index factor (mo_transform_factor: north_star's density-fitted 3-index route) against the four quarter transformations of
    the packed block: the same energies, densities, amplitudes and relaxed densities; the route follows qemb_frag_mo_route / the cost rule; new ERIs drop the factor."""
    from helpers import synthetic_fragment_factor
    from quemb_amd._lib import QembError
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    for n, o, nf, naux in cases:
        h, e1, Bp = synthetic_fragment_factor(n, o, 300 + n, naux=naux or None)
        rng = np.random.default_rng(n)
        h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
        veff0 = rng.standard_normal((n, n)); veff0 = veff0 + veff0.T
        s4 = eri.pack_s4(e1)
        assert np.abs(Bp.T @ Bp - s4).max() < 1e-12
        fr = DeviceFragment(n, nf, lib=lib)
        fr.set_eri_s4(s4)
        fr.set_energy_data(h1, veff0, None, 0.5, [0, 1])
        assert fr.mo_route_used() == (False, 0)
        outs = {}
        for relax in (0, 1):
            opts = default_opts(lib, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9, relax_density=relax, lambda_conv_tol=1e-11)
            fr.set_eri_s4(s4)                                  # (drops the factor)
            fr.set_mo_route(-1)
            outs[relax, "four"] = fr.solve(o, h, opts=opts, eeval=True, want_t2=True)
            assert fr.mo_route_used() == (False, 0)
            fr.set_df_factor(Bp)
            fr.set_mo_route(1)
            outs[relax, "factor"] = fr.solve(o, h, opts=opts, eeval=True, want_t2=True)
            assert fr.mo_route_used() == (o < n, Bp.shape[0])          # (no virtual orbitals: no MO integrals at all)
            a, b = outs[relax, "four"], outs[relax, "factor"]
            assert a["n_iter"] == b["n_iter"] and a["lambda_iters"] == b["lambda_iters"]
            for k in ("e_corr_mo", "e_scf", "ebe_hf"):
                assert abs(a[k] - b[k]) < tol, (n, relax, k, a[k], b[k])
            for k in ("e_frag", "rdm1_emb", "rdm1_mo", "mo_energy"):
                assert np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() < tol, (n, relax, k)
            if o < n:
                # the same SCF ran before both: identical orbitals, so the amplitudes compare element by element
                assert np.array_equal(a["mo_coeff"], b["mo_coeff"])
                assert np.abs(a["t1"] - b["t1"]).max() < tol and np.abs(a["t2"] - b["t2"]).max() < tol
        # by cost: the default follows mo_factor_route_pays (naux <= 8 n: every case but the last)
        fr.set_mo_route(-1)
        fr.solve(o, h, opts=opts, eeval=True)
        assert fr.mo_route_used() == (Bp.shape[0] <= 8 * n and o < n, Bp.shape[0])
        fr.set_mo_route(0)
        fr.solve(o, h, opts=opts, eeval=True)
        assert fr.mo_route_used() == (False, Bp.shape[0])
        # CPHF response (HF Jacobian of the QN optimiser) goes through the same integrals
        if 0 < o < n:
            v1 = rng.standard_normal((2, n, n)); v1 = v1 + v1.transpose(0, 2, 1)
            fr.set_mo_route(1); d_fac = fr.cphf(o, h, v1)
            fr.set_mo_route(0); d_four = fr.cphf(o, h, v1)
            assert np.abs(d_fac - d_four).max() < 1e-9
        fr.set_eri_s4(s4)
        assert fr.mo_route_used()[1] == 0
        fr.set_mo_route(1)
        with pytest.raises(QembError):
            fr.solve(o, h, opts=opts, eeval=True)             # the factor route without a factor is an error, not a silent other route
        with pytest.raises(ValueError):
            fr.set_df_factor(Bp[:, :-1])
        with pytest.raises(QembError, match="not the factor of these ERIs"):
            fr.set_df_factor(1.01 * Bp)                         # a factor of other integrals is refused, and none is kept
        assert fr.mo_route_used()[1] == 0
        if Bp.shape[1] > 17:
            # ... also one that agrees with the block in its leading 16 x 16 corner and differs elsewhere (a stale / truncated factor, a fragment sharing
            # its first pairs): the whole block is probed (round-4 review; the corner-only check accepted this)
            Bbad = Bp.copy(); Bbad[:, 16:] *= 1.001
            with pytest.raises(QembError, match="not the factor of these ERIs"):
                fr.set_df_factor(Bbad)
            assert fr.mo_route_used()[1] == 0
            fr.set_df_factor(Bp)                                # the right one is still accepted afterwards
            assert fr.mo_route_used()[1] == Bp.shape[0]


def test_factor_route_equals_four_index(hlib):
    check_factor_route_equals_four_index(hlib)


def check_fragment_living_on_its_factor(lib, cases=((8, 3, 3, 20), (11, 4, 4, 0), (9, 2, 3, 31), (7, 7, 2, 12)), tol=2e-10, jk_tol=1e-11):
    """A fragment that keeps its 3-index factor ALONE (qemb_frag_set_df_only: no 4-fold packed block resident, round-4 review item 3): J / K against the oracle's get_jk
    (molbe/helper.py:28-69) for a general matrix, the whole solve (fragment RHF with J / K from the factor -> MO integrals -> CCSD -> densities -> energies, unrelaxed and
    relaxed, CPHF) against the same fragment with the block resident, the block on demand (get_eri_s4 == B^T B), the forced four-index route on a transient block,
    and the resident bytes."""
    from helpers import synthetic_fragment_factor
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    for n, o, nf, naux in cases:
        h, e1, Bp = synthetic_fragment_factor(n, o, 500 + n, naux=naux or None)
        rng = np.random.default_rng(n)
        h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
        veff0 = rng.standard_normal((n, n)); veff0 = veff0 + veff0.T
        veff = rng.standard_normal((n, n)); veff = veff + veff.T
        s4 = eri.pack_s4(e1)
        npair = n * (n + 1) // 2
        ref = DeviceFragment(n, nf, lib=lib); ref.set_eri_s4(s4); ref.set_df_factor(Bp); ref.set_mo_route(1)
        fr = DeviceFragment(n, nf, lib=lib); fr.set_df_only(Bp)
        for f in (ref, fr):
            f.set_energy_data(h1, veff0, veff, 0.5, [0, 1])
        assert fr.resident_bytes() == 8 * Bp.size and ref.resident_bytes() == 8 * (Bp.size + npair * npair)
        # J / K of a general (non-symmetric, indefinite) matrix and of a density
        for P in (rng.standard_normal((n, n)), (lambda X: X @ X.T)(rng.standard_normal((n, max(o, 1))))):
            J, K = fr.jk(P)
            Jr, Kr = scf.get_jk(e1, P)
            assert np.abs(J - Jr).max() < jk_tol * max(1.0, np.abs(Jr).max()) and np.abs(K - Kr).max() < jk_tol * max(1.0, np.abs(Kr).max()), (n, np.abs(J - Jr).max(), np.abs(K - Kr).max())
        assert np.abs(fr.get_eri_s4() - s4).max() < 1e-12
        for relax in (0, 1):
            opts = default_opts(lib, relax_density=relax)
            a = ref.solve(o, h, opts=opts, eeval=True, want_t2=True)
            b = fr.solve(o, h, opts=opts, eeval=True, want_t2=True)
            assert fr.mo_route_used() == (o < n, Bp.shape[0])
            assert a["n_iter"] == b["n_iter"] and a["scf_cycles"] == b["scf_cycles"] and a["lambda_iters"] == b["lambda_iters"]
            for k in ("e_corr_mo", "e_scf", "ebe_hf"):
                assert abs(a[k] - b[k]) < tol, (n, relax, k, a[k], b[k])
            for k in ("e_frag", "rdm1_emb", "rdm1_mo", "mo_energy"):
                assert np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() < tol, (n, relax, k)
            if o < n:
                # amplitudes through phase-invariant contractions (the two SCFs took J / K by different routes: orbitals agree to rounding, signs may not)
                ta = a["mo_coeff"][:, :o] @ a["t1"] @ a["mo_coeff"][:, o:].T
                tb = b["mo_coeff"][:, :o] @ b["t1"] @ b["mo_coeff"][:, o:].T
                assert np.abs(ta - tb).max() < 10 * tol
            # warm second solve from the previous density (the path of a BE sweep)
            dm0 = 2.0 * b["mo_coeff"][:, :o] @ b["mo_coeff"][:, :o].T
            b2 = fr.solve(o, h, dm0=dm0, opts=opts, eeval=True)
            # (another starting point: the two solves agree to the convergence thresholds of the SCF / CCSD, not to rounding)
            assert abs(b2["e_corr_mo"] - b["e_corr_mo"]) < 1e-9 and np.abs(b2["rdm1_emb"] - b["rdm1_emb"]).max() < 1e-8
        # the four-index route forced on a factor-only fragment: the block is a transient of the solve
        fr.set_mo_route(0)
        c = fr.solve(o, h, opts=default_opts(lib), eeval=True)
        assert fr.mo_route_used() == (False, Bp.shape[0]) and fr.resident_bytes() < 8 * (Bp.size + npair * npair)
        a0 = ref.solve(o, h, opts=default_opts(lib), eeval=True)
        assert abs(c["e_corr_mo"] - a0["e_corr_mo"]) < tol and np.abs(c["e_frag"] - a0["e_frag"]).max() < tol
        fr.set_mo_route(-1)
        if 0 < o < n:
            v1 = rng.standard_normal((2, n, n)); v1 = v1 + v1.transpose(0, 2, 1)
            assert np.abs(fr.cphf(o, h, v1) - ref.cphf(o, h, v1)).max() < 1e-9
        sc = fr.scf(o, h)
        sr = ref.scf(o, h)
        assert abs(sc["e_scf"] - sr["e_scf"]) < tol and np.abs(sc["J"] - sr["J"]).max() < tol and np.abs(sc["K"] - sr["K"]).max() < tol
        fr.set_eri_s4(s4)                                   # new ERIs end the mode
        assert fr.mo_route_used()[1] == 0 and fr.resident_bytes() >= 8 * npair * npair


def test_fragment_living_on_its_factor(hlib):
    check_fragment_living_on_its_factor(hlib)


def check_wide_diis_space_takes_the_general_path(lib):
    """More stored DIIS vectors than the fused end-of-iteration launches take (eight): the pass-by-pass path (lincomb / dot_many / outer4 / reductions, stream waits)
    runs instead and converges to the same amplitudes; the iteration counts differ (another extrapolation space), the fixed point does not."""
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o, nf = 9, 3, 3
    h, e1, h1, veff0, veff = _problem(n, o, nf, 77)
    outs = []
    for space in (6, 10):
        fr = DeviceFragment(n, nf, lib=lib)
        fr.set_eri_s4(eri.pack_s4(e1))
        fr.set_energy_data(h1, veff0, veff, 0.75, [0, 1])
        opts = default_opts(lib, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9, cc_diis_space=space)
        outs.append(fr.solve(o, h, opts=opts, eeval=True))
    a, b = outs
    assert abs(a["e_corr_mo"] - b["e_corr_mo"]) < 1e-10 and np.abs(a["rdm1_emb"] - b["rdm1_emb"]).max() < 1e-8
    assert np.allclose(a["e_frag"], b["e_frag"], atol=1e-9)


def test_wide_diis_space_takes_the_general_path(hlib):
    check_wide_diis_space_takes_the_general_path(hlib)


def test_single_update_amps_matches_oracle(hlib):
    """One un-extrapolated amplitude update from the MP2 guess, then DIIS-free convergence to the same energy."""
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o, nf = 7, 3, 2
    h, e1, *_ = _problem(n, o, nf, 99)
    fr = DeviceFragment(n, nf, lib=hlib)
    fr.set_eri_s4(eri.pack_s4(e1))
    opts = default_opts(hlib, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9)
    fr.prepare_ccsd(o, h, opts=opts)
    mf = scf.rhf(h, e1, o, conv_tol=1e-13, conv_tol_grad=1e-9)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    t1, t2 = ccsd.init_amps(eris)
    for it in range(3):   # first iterations carry no DIIS extrapolation difference only at it == 0
        e_g, nt_g = fr.ccsd_iterate(1)
        t1n, t2n = ccsd.update_amps(t1, t2, eris)
        if it == 0:
            e_o = ccsd.energy(t1n, t2n, eris)
            nt_o = np.sqrt(np.linalg.norm(t1n - t1) ** 2 + np.linalg.norm(t2n - t2) ** 2)
            assert abs(e_g - e_o) < 1e-12 and abs(nt_g - nt_o) < 1e-12
        t1, t2 = t1n, t2n


def test_update_amps_with_split_k_slabs(hlib):
    """n_virt = 64: the packed pair index is 2080 long, past the threshold from which the pair products are split over K and leave their partial
    products in slabs that the scatter kernels add up (no reduction pass) -- one update and its energy against the oracle."""
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o, nf = 67, 3, 2
    h, e1, *_ = _problem(n, o, nf, 5)
    fr = DeviceFragment(n, nf, lib=hlib)
    fr.set_eri_s4(eri.pack_s4(e1))
    opts = default_opts(hlib, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9)
    fr.prepare_ccsd(o, h, opts=opts)
    mf = scf.rhf(h, e1, o, conv_tol=1e-13, conv_tol_grad=1e-9)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    t1, t2 = ccsd.init_amps(eris)
    e_g, nt_g = fr.ccsd_iterate(1)
    t1n, t2n = ccsd.update_amps(t1, t2, eris)
    e_o = ccsd.energy(t1n, t2n, eris)
    nt_o = np.sqrt(np.linalg.norm(t1n - t1) ** 2 + np.linalg.norm(t2n - t2) ** 2)
    assert abs(e_g - e_o) < 1e-11 and abs(nt_g - nt_o) < 1e-11


def test_warm_start_and_errors(hlib):
    from quemb_amd._lib import QembError
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o, nf = 6, 2, 2
    h, e1, h1, veff0, veff = _problem(n, o, nf, 5)
    fr = DeviceFragment(n, nf, lib=hlib)
    with pytest.raises(QembError):
        fr.solve(o, h)                       # ERIs not set
    fr.set_eri_s4(eri.pack_s4(e1))
    with pytest.raises(QembError):
        fr.solve(o, h, eeval=True)           # energy data not set
    with pytest.raises(ValueError):
        fr.set_eri_s4(np.zeros((3, 3)))
    a = fr.solve(o, h, eeval=False)
    b = fr.solve(o, h, opts=default_opts(hlib, warm_start=1), eeval=False)
    assert abs(a["e_corr_mo"] - b["e_corr_mo"]) < 1e-9 and b["n_iter"] <= 3
    with pytest.raises(QembError):
        fr.solve(o, h, opts=default_opts(hlib, cc_max_cycle=1), eeval=False)   # non-convergence is an error
    check_non_strict_convergence(hlib)


def check_non_strict_convergence(lib):
    """strict_convergence = 0: the reference's behaviour (PySCF warns and carries on, helper.py:128-149, solver.py:905-912) -- the
    unconverged CCSD / Lambda / RHF state is returned with a ConvergenceWarning instead of an error."""
    import warnings
    from quemb_amd._lib import ConvergenceWarning
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o, nf = 6, 2, 2
    h, e1, h1, veff0, veff = _problem(n, o, nf, 5)
    fr = DeviceFragment(n, nf, lib=lib)
    fr.set_eri_s4(eri.pack_s4(e1))
    fr.set_energy_data(h1, veff0, veff, 1.0, [0])
    ref = fr.solve(o, h, eeval=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fr.solve(o, h, opts=default_opts(lib, cc_max_cycle=2, strict_convergence=0), eeval=True)
    assert any(issubclass(x.category, ConvergenceWarning) and "CCSD did not converge" in str(x.message) for x in w)
    assert out["n_iter"] == 2 and 1e-9 < abs(out["e_corr_mo"] - ref["e_corr_mo"]) < 0.1 * abs(ref["e_corr_mo"])
    assert np.abs(out["rdm1_emb"] - ref["rdm1_emb"]).max() < 0.1 and np.isfinite(np.asarray(out["e_frag"])).all()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fr.solve(o, h, opts=default_opts(lib, relax_density=1, lambda_max_cycle=1, strict_convergence=0), eeval=True)
    assert any("Lambda" in str(x.message) for x in w) and abs(out["e_corr_mo"] - ref["e_corr_mo"]) < 1e-9
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fr.solve(o, h, opts=default_opts(lib, scf_max_cycle=1, strict_convergence=0), eeval=False)
    assert any(issubclass(x.category, ConvergenceWarning) for x in w)
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # a converged solve raises no warning
        fr.solve(o, h, opts=default_opts(lib, strict_convergence=0), eeval=True)
    fr.free()


@pytest.mark.parametrize("n,o,nf,cen", [(6, 2, 3, [0, 1]), (8, 3, 3, [1]), (7, 4, 2, [0])])
def test_relaxed_density_fragment_solve_matches_oracle(hlib, n, o, nf, cen):
    """relax_density=True (solve_ccsd(relax=True), molbe/solver.py:925-939): Lambda equations, response 1-RDM and the
    relaxed with_dm1=False 2-RDM contracted into the fragment energy -- against oracle/qemb_oracle/ccsd_lambda.py."""
    from qemb_oracle import ccsd_lambda
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    h, e1, h1, veff0, veff = _problem(n, o, nf, 70 + n)
    s4 = eri.pack_s4(e1)
    fr = DeviceFragment(n, nf, lib=hlib)
    fr.set_eri_s4(s4)
    fr.set_energy_data(h1, veff0, veff, 0.75, cen)
    opts = default_opts(hlib, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11, scf_conv_tol=1e-13, scf_conv_tol_grad=1e-9,
                        relax_density=1, lambda_conv_tol=1e-11)
    out = fr.solve(o, h, opts=opts, eeval=True)
    assert out["lambda_iters"] > 1
    mf = scf.rhf(h, e1, o, conv_tol=1e-13, conv_tol_grad=1e-9)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, ecc, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-13, conv_tol_normt=1e-11)
    z1, z2, nit, lag = ccsd_lambda.solve_lambda(t1, t2, eris, conv_tol=1e-12)
    dm1, _ = ccsd_lambda.response_densities(lag, z1, z2)
    g2 = ccsd_lambda.make_rdm2_relaxed(lag, z1, z2)
    C = mf["mo_coeff"]
    assert np.abs(out["rdm1_emb"] - 0.5 * C @ dm1 @ C.T).max() < 1e-8
    assert abs(np.trace(out["rdm1_mo"]) - 2 * o) < 1e-9                  # the response density keeps the electron count
    TA = np.zeros((n + 2, n))
    e_ref = be.get_frag_energy(C, o, nf, (0.75, cen), TA, h1, dm1, g2, s4, veff0, None, True)
    assert np.allclose(out["e_frag"], e_ref, atol=1e-8), (out["e_frag"], e_ref)
    # and it differs from the unrelaxed answer (the test would otherwise not see the Lambda part)
    out0 = fr.solve(o, h, opts=default_opts(hlib, cc_conv_tol=1e-13, cc_conv_tol_normt=1e-11), eeval=True)
    assert out0["lambda_iters"] == 0 and abs(out0["e_frag"][1] - out["e_frag"][1]) > 1e-6


def test_lambda_nonconvergence_is_an_error(hlib):
    """Non-convergence is reported, not returned silently (status -4 like the amplitude equations)."""
    from quemb_amd._lib import QembError
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    h, e1, h1, veff0, veff = _problem(6, 2, 3, 5)
    fr = DeviceFragment(6, 3, lib=hlib)
    fr.set_eri_s4(eri.pack_s4(e1))
    with pytest.raises(QembError, match="Lambda"):
        fr.solve(2, h, opts=default_opts(hlib, relax_density=1, lambda_max_cycle=1), eeval=False)
    out = fr.solve(2, h, opts=default_opts(hlib, relax_density=1), eeval=False)     # and the handle is still usable
    assert out["lambda_iters"] > 1


def check_fragment_without_virtual_orbitals(lib):
    """nsocc == n (solver.py:829-946 on a mean field with no virtuals: PySCF's CCSD has empty amplitudes, E_corr = 0): the sweep body
    returns the mean-field results -- E_scf of the full density 2 I, rdm1 = 2 I (MO) / I (embedding), zero correlation contributions to
    the fragment energies, the fragment HF energy of update_ebe_hf -- with and without relax_density, and a later solve with virtuals on
    the same fragment object is unaffected."""
    from qemb_oracle import scf as oscf
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    for n in (1, 3, 4):
        h, e1 = synthetic_fragment(n, max(1, n - 1), 40 + n)
        vj, vk = oscf.get_jk(e1, 2.0 * np.eye(n))
        e_hf = float(np.sum((h + 0.5 * (vj - 0.5 * vk)) * 2.0 * np.eye(n)))
        fr = DeviceFragment(n, 1, lib=lib)
        fr.set_eri_s4(eri.pack_s4(e1))
        fr.set_energy_data(h, 0.3 * h, 0.2 * h, 1.0, [0])
        for relax in (0, 1):
            out = fr.solve(n, h, opts=default_opts(lib, relax_density=relax), eeval=True, want_t2=True)
            assert out["e_corr_mo"] == 0.0 and out["n_iter"] == 0 and out["t1"].size == 0 and out["t2"].size == 0
            assert abs(out["e_scf"] - e_hf) < 1e-10
            assert np.abs(out["rdm1_mo"] - 2.0 * np.eye(n)).max() < 1e-14 and np.abs(out["rdm1_emb"] - np.eye(n)).max() < 1e-12
            assert np.abs(np.asarray(out["e_frag"])).max() == 0.0
            # update_ebe_hf (pfrag.py:327-400) for the centre 0 with D = Co Co^T = I:  2 h1_00 + veff_00 + (J - K/2)_00 of dm = 2 I
            e_be = 2.0 * h[0, 0] + 0.2 * h[0, 0] + vj[0, 0] - 0.5 * vk[0, 0]
            assert abs(out["ebe_hf"] - e_be) < 1e-10, (out["ebe_hf"], e_be)
        if n > 1:
            ok = fr.solve(n - 1, h, eeval=False)
            assert ok["n_iter"] > 0 or n == 2
        fr.free()


def test_fragment_without_virtual_orbitals(hlib):
    check_fragment_without_virtual_orbitals(hlib)


def test_pair_gemm_tile_choice(hlib):
    """the row tile of the pp-ladder / tau-dressing products is chosen by estimated time, not by least padding alone: the slow
    single-column wave tiles (cfg 11 / 12) win only where they save much more than their rate costs."""
    import ctypes as C
    npair = lambda x: x * (x + 1) // 2

    def choice(rows, cols=20100):
        c, k = C.c_int(), C.c_int()
        assert hlib.qemb_pair_gemm_choice(rows, cols, C.byref(c), C.byref(k)) == 0
        return c.value, k.value
    assert choice(npair(20))[0] == 13 and choice(190)[0] == 15            # the benchmarked fragment (o = 20): 224- and 192-row tiles
    assert choice(npair(30))[0] == 35                                      # 465 rows: three 160-row tiles
    assert choice(npair(40))[0] in (13, 4)                                 # 820 rows: 896 padded rows on a fast tile, NOT thirteen 64-row tiles
    assert choice(npair(40) - 40)[0] in (13, 15, 35, 4)                    # 780 antisymmetric rows
    assert choice(36)[0] == 38 and choice(45)[0] == 38 and choice(100)[0] in (4, 11)      # n_occ = 9: 45 / 36 pair rows on the 48 x 128 tile (round 5), not on 80 rows
    assert choice(49)[0] in (12, 36)
    assert choice(78)[0] == 36 and choice(66)[0] == 36                         # n_occ = 12 (mid-size fragments): 78 / 66 pair rows on the 80-row tile, not on 128 rows
    assert choice(210, cols=1000) == (-1, 0)                               # few columns: the dispatcher's own choice
    for rows in (28, 105, 190, 210, 465, 820, 1275):
        cfg, ks = choice(rows)
        assert 1 <= ks <= 8


def check_solve_batch_equals_one_by_one(lib, sizes=((8, 3, 3), (10, 4, 4), (7, 2, 2), (9, 3, 3), (6, 6, 2)), expect_grouped=False):
    """qemb_frag_solve_batch (fragment phases per stream, CCSD iterations in lock step) == qemb_frag_solve fragment by fragment, bit for bit:
    fragments of different sizes (they converge in different iterations and drop out of the lock step one by one), one without virtual
    orbitals, energies on.  On the GPU the lock-step iterations must have issued grouped launches."""
    from quemb_amd.fragsolver import DeviceFragment, default_opts, solve_batch
    frs, hs, outs_ref = [], [], []
    opts = default_opts(lib)
    for k, (n, o, nf) in enumerate(sizes):
        h, e1, h1, veff0, veff = _problem(n, o, nf, 900 + k)
        fr = DeviceFragment(n, nf, lib=lib)
        fr.set_eri_s4(eri.pack_s4(e1))
        if k % 2:       # a mixed batch: every second fragment holds its 3-index factor and forms its MO integrals from it (both ways of solving it do)
            from helpers import synthetic_fragment_factor
            fr.set_df_factor(synthetic_fragment_factor(n, o, 900 + k)[2])
        fr.set_energy_data(h1, veff0, veff, 1.0, list(range(nf)))
        frs.append(fr); hs.append(h)
        outs_ref.append(fr.solve(o, h, None, opts=opts, eeval=True, want_t2=True))
        assert fr.mo_route_used()[0] == bool(k % 2 and o < n)
    stats = {}
    outs = solve_batch(frs, [s[1] for s in sizes], hs, None, opts=opts, eeval=True, want_t2=True, stats=stats)
    assert len({o["n_iter"] for o in outs_ref}) > 1          # the fragments do not all converge together
    for a, b in zip(outs, outs_ref):
        assert a["n_iter"] == b["n_iter"] and a["scf_cycles"] == b["scf_cycles"]
        for key in ("e_corr_mo", "e_scf", "ebe_hf"):
            assert a[key] == b[key], key
        for key in ("e_frag", "mo_coeff", "mo_energy", "rdm1_emb", "rdm1_mo", "t1", "t2"):
            assert np.array_equal(a[key], b[key]), key
    if expect_grouped:
        assert stats["merged_runs"] > 0 and stats["grouped_launches"] > 0 and stats["launches"] < stats["operations"]
        assert stats["max_group"] >= 2
    # a second batch over the same handles (new tapes, new plans) and a batch of one
    outs2 = solve_batch(frs[:2], [s[1] for s in sizes[:2]], hs[:2], None, opts=opts, eeval=True)
    assert outs2[0]["e_corr_mo"] == outs_ref[0]["e_corr_mo"] and outs2[1]["e_corr_mo"] == outs_ref[1]["e_corr_mo"]
    one = solve_batch(frs[1:2], [sizes[1][1]], hs[1:2], None, opts=opts, eeval=True)
    assert one[0]["e_corr_mo"] == outs_ref[1]["e_corr_mo"]
    from quemb_amd._lib import QembError
    with pytest.raises(QembError):
        solve_batch([frs[0], frs[0]], [3, 3], [hs[0], hs[0]], None, opts=opts)       # the same fragment twice
    with pytest.raises(QembError):
        solve_batch(frs[:2], [3, 4], hs[:2], None, opts=default_opts(lib, cc_max_cycle=1))      # non-convergence is an error for the whole batch
    for fr in frs:
        fr.free()


def test_solve_batch_equals_one_by_one(hlib):
    check_solve_batch_equals_one_by_one(hlib)


def test_fused_scf_ops_on_the_mock(hlib):
    """the fused steps of the SCF cycle of small fragments through the C ABI of the mock build (the GPU test of the same name runs the HIP kernels)"""
    from test_gpu_ops import check_fused_scf_ops
    check_fused_scf_ops(hlib, (2, 7, 24, 41))


def test_small_copies_on_the_mock(hlib):
    from test_gpu_ops import check_small_copies
    check_small_copies(hlib)
