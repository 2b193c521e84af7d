"""Gamma-point periodic direct DF transform through the C ABI (qemb_df_create_pbc / alloc_ints / add_pw_block / add_rs_block /
transform) against the reference's own outputs (tests/golden/kbe_df.npz) -- on the scalar mock here, on the HIP library under -m gpu
(tests/test_gpu_be.py imports `check_periodic_df`)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
from kbe_df_source import CASES, SyntheticGammaSource, fragment_TAs, make_case  # noqa: E402

G = np.load(Path(__file__).parent / "golden" / "kbe_df.npz")


@pytest.fixture(scope="module")
def hlib():
    import build as hc_build
    from quemb_amd import _lib
    return _lib.declare(C.CDLL(str(hc_build.build())))


class _Frag:
    def __init__(self, TA):
        self.TA, self.dev = TA, None


def check_periodic_df(lib, tol=1e-10):
    from quemb_amd import eri_transform as et
    from quemb_amd import kbe_eri_onthefly as keo
    for name in CASES:
        src, TAs, raises = make_case(name)
        df = et.DFContext.periodic(src.j2c(), lib=lib)
        assert int(df.ischol) == int(G[f"{name}/ischol"]), name            # kbe/eri_onthefly.py:19-45
        df.free()
        frs = [_Frag(TA) for TA in TAs]
        for steps in ((4, 3), (1000, 1000)):
            if raises:
                with pytest.raises(ValueError, match="Imaginary part of ERI is larger than 1e-6 for frag #0"):
                    keo.integral_direct_DF(src, frs, *steps, lib=lib, want_host=True)
                continue
            eris = keo.integral_direct_DF(src, frs, *steps, lib=lib, want_host=True)
            for i, e in enumerate(eris):
                ref = G[f"{name}/eri{i}"]
                assert np.abs(e - ref).max() < tol * max(1.0, np.abs(ref).max()), (name, i, steps, np.abs(e - ref).max())


def check_periodic_df_argument_errors(lib):
    from quemb_amd import eri_transform as et
    from quemb_amd._lib import QembError
    src, TAs, _ = make_case("pd")
    df = et.DFContext.periodic(src.j2c(), lib=lib)
    with pytest.raises(ValueError):
        df.add_pw_block(np.zeros((src.naux, 2), complex), np.zeros((2, src.nao, src.nao), complex))     # alloc_ints first
    df.alloc_ints(src.nao)
    with pytest.raises(QembError):
        df.add_rs_block(src.naux - 1, np.zeros((2, src.nao, src.nao)))                                   # rows past naux
    with pytest.raises(QembError):
        df.select_part(3)
    df.free()
    plain = et.DFContext(j2c=np.eye(4), lib=lib)
    with pytest.raises(QembError):
        plain.imag_absmax()                                                                               # no plane-wave accumulation
    plain.free()


def test_periodic_df_matches_reference_outputs_on_the_mock(hlib):
    check_periodic_df(hlib)


def test_periodic_df_argument_errors_on_the_mock(hlib):
    check_periodic_df_argument_errors(hlib)


def big_case():
    """sizes past one GEMM tile in every dimension, several plane-wave and auxiliary blocks"""
    src = SyntheticGammaSource(nao=40, naux=90, nhalf=70, seed=77, indefinite=True)
    return src, fragment_TAs(40, (22, 17), 78)


def test_periodic_df_beyond_one_tile_on_the_mock(hlib):
    from qemb_oracle import kbe_df
    from quemb_amd import kbe_eri_onthefly as keo
    src, TAs = big_case()
    ref, ischol = kbe_df.integral_direct_DF(src, TAs, 16, 7)
    assert not ischol
    got = keo.integral_direct_DF(src, [_Frag(TA) for TA in TAs], pw_step=32, aux_step=25, lib=hlib, want_host=True)
    for e, r in zip(got, ref):
        assert np.abs(e - r).max() < 1e-9 * np.abs(r).max()
