"""CPU: the AO-screening restatement (oracle/qemb_oracle/sparse_df.py) against an independent grid integral, the device quadrature
kernel (scalar mock here, HIP under -m gpu) against it, and the geometry-only semi-sparse DF pipeline against the dense DF one."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest
from scipy.special import roots_hermite

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
from qemb_oracle import sparse_df as osd  # noqa: E402

SPD = {"C": [(0, [2.9, 0.68, 0.22], [0.2, 0.5, 0.4]), (1, [1.1, 0.3], [0.5, 0.6]), (2, [0.8], [1.0])], "H": [(0, [1.2, 0.3], [0.4, 0.7])]}


@pytest.fixture(scope="module")
def hlib():
    import build as hc_build
    from quemb_amd import _lib
    return _lib.declare(C.CDLL(str(hc_build.build())))


@pytest.mark.parametrize("li,lj", [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (4, 3)])
def test_primitive_abs_overlap_against_a_grid_integral(li, lj):
    r, w = roots_hermite(500)
    Ra, Rb = [0.1, -0.2, 0.0], [0.5, 0.3, -0.4]
    q = osd.primitive_abs_overlap(li, lj, 0.8, 1.1, Ra, Rb, r, w)
    g = osd.abs_overlap_grid(li, lj, 0.8, 1.1, Ra, Rb, n=2401, box=8.0)
    # the reference's 500-point Gauss-Hermite rule is itself an approximation here: |x - A|^p has a kink at the centre, which a
    # polynomial rule integrates to ~1e-3 only (the fine trapezoid grid is the more accurate of the two)
    assert np.abs(q - g).max() < 2e-3 * np.abs(g).max()
    assert (q > 0).all()


def check_abs_overlap_and_reachability(lib):
    from quemb_amd.eri_sparse_DF import _get_AO_per_AO, _primitive_shells, approx_S_abs
    from quemb_amd.integrals import Mole
    for atoms, basis in (([["C", (0, 0, 0)], ["H", (0.9, 0.3, -0.2)], ["H", (-0.5, 0.8, 0.4)], ["C", (3.5, 0.2, 0.1)]], SPD),
                         ([["H", (0, 0, float(i))] for i in range(6)], "sto-3g")):
        mol = Mole(atoms, basis=basis)
        S = approx_S_abs(mol, lib=lib)
        ls, exps, xyz, _, _, A = _primitive_shells(mol)
        So = osd.approx_S_abs([(int(l), float(e), x) for l, e, x in zip(ls, exps, xyz)], A @ np.abs(mol.c2s))
        assert np.abs(S - So).max() < 1e-12
        assert np.abs(np.diag(S) - 1).max() < 1e-12 and (S > 0).all() and np.abs(S - S.T).max() < 1e-14
        rng = np.random.default_rng(3)
        TA = rng.standard_normal((mol.nao, 4)) * (np.arange(mol.nao)[:, None] < 5)
        for eps in (1e-10, 0.2, 0.6):
            for ta in (None, TA):
                got, ref = _get_AO_per_AO(S, eps, ta, lib=lib), osd.get_AO_per_AO(S, eps, ta)
                assert got == ref


def test_device_abs_overlap_and_reachability_match_the_restatement(hlib):
    check_abs_overlap_and_reachability(hlib)


def check_sparse_df_from_geometry(lib, atoms=None, frag_key="test_autogen_h_linear_be2", tol=1e-9):
    """BE(int_transform="sparse-DF-hip" | "on-fly-sparse-DF-hip" | "int-direct-DF-hip", auxbasis=...) from the geometry alone: the
    semi-sparse pipeline (AO screening 1e-10, MO screening off) == the dense DF pipeline with the same auxiliary basis, and both sit
    within the fitting error of the in-core result, which shrinks when higher auxiliary angular momenta are added."""
    from helpers import GOLDEN
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole, etb_auxbasis
    from quemb_amd.mbe import BE
    mol = Mole(atoms or [["H", (0.0, 0.0, float(i))] for i in range(8)])
    mf = RHF(mol); mf.kernel()
    fobj = FragPart.from_json(GOLDEN / "fragmentation.json", frag_key)
    e_in = BE(mf, fobj, lib=lib, distribute=False).oneshot()[0]
    errs = []
    for lmax in (0, 2):
        aux = etb_auxbasis(mol, beta=1.8, lmax_by_symbol={"H": lmax, "C": lmax + 1})
        e_dense = BE(mf, fobj, lib=lib, distribute=False, int_transform="int-direct-DF-hip", auxbasis=aux).oneshot()[0]
        be_sp = BE(mf, fobj, lib=lib, distribute=False, int_transform="sparse-DF-hip", auxbasis=aux, MO_coeff_epsilon=0.0)
        e_sp = be_sp.oneshot()[0]
        e_fly = BE(mf, fobj, lib=lib, distribute=False, int_transform="on-fly-sparse-DF-hip", auxbasis=aux, MO_coeff_epsilon=0.0).oneshot()[0]
        assert abs(e_sp - e_dense) < tol and abs(e_fly - e_dense) < tol, (e_sp, e_fly, e_dense)
        assert be_sp.df_stats["n_unique"] <= be_sp.df_stats["n_pairs_dense"]
        errs.append(abs(e_dense - e_in))
    assert errs[1] < errs[0] and errs[1] < 1e-3, errs
    # the reference's default thresholds (MO_coeff_epsilon 1e-5): a screened result within the screening error
    e_scr = BE(mf, fobj, lib=lib, distribute=False, int_transform="sparse-DF-hip", auxbasis=aux).oneshot()[0]
    assert abs(e_scr - e_dense) < 1e-5
    with pytest.raises(ValueError):
        BE(mf, fobj, lib=lib, distribute=False, int_transform="sparse-DF-hip")
    return errs


def test_sparse_df_from_geometry_h8(hlib):
    check_sparse_df_from_geometry(hlib)
