"""Row a4 (molbe/eri_onthefly.py:45-145) against the reference's OWN outputs: tests/golden/df.npz was written by
tests/golden/make_golden_df.py, which runs `quemb.molbe.eri_onthefly.integral_direct_DF` on the synthetic integrals of tests/df_source.py.
Here: the oracle restatement (1e-11) and the C ABI (qemb_df_create / set_ints / transform) on the scalar mock; the HIP library runs the
same check under -m gpu (tests/test_gpu_be.py imports `check_df_golden`)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
from df_source import CASES, make_case  # noqa: E402

G = np.load(Path(__file__).parent / "golden" / "df.npz")


def test_oracle_matches_reference_integral_direct_DF():
    from qemb_oracle import eri as oeri
    for name in CASES:
        src, TAs, _ = make_case(name)
        for i, TA in enumerate(TAs):
            ref = G[f"{name}/eri{i}"]
            got = oeri.integral_direct_DF(src.pqL, src.j2c, TA)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), (name, i)


def check_df_golden(lib, tol=1e-10):
    """every storage layout qemb_df_set_ints accepts, each fragment of each case, against the reference's dataset f{i}"""
    from quemb_amd import eri_transform as et
    for name in CASES:
        src, TAs, _ = make_case(name)
        N = src.nao
        il = np.tril_indices(N)
        layouts = {"pqL": src.pqL, "Lpq": np.ascontiguousarray(src.pqL.transpose(2, 0, 1)),
                   "packed": np.ascontiguousarray(src.pqL.transpose(2, 0, 1)[:, il[0], il[1]])}
        for layout, ints in layouts.items():
            df = et.DFContext(j2c=src.j2c, lib=lib)
            df.set_ints(ints, N, layout)
            for i, TA in enumerate(TAs):
                ref = G[f"{name}/eri{i}"]
                out = df.transform(TA, want_host=True)
                assert np.abs(out - ref).max() < tol * max(1.0, np.abs(ref).max()), (name, layout, i, np.abs(out - ref).max())
            df.free()
        # delivered straight into a fragment the block arrives WITH the fitted factor bb it was formed from (eri_onthefly.py:141-143): bb^T bb is the
        # block, and a solve through the factor route gives what the four-index route gives
        from quemb_amd.fragsolver import DeviceFragment, default_opts
        df = et.DFContext(j2c=src.j2c, lib=lib)
        df.set_ints(layouts["packed"], N, "packed")
        n = TAs[0].shape[1]
        fr = DeviceFragment(n, min(2, n), lib=lib)
        out = df.transform(TAs[0], frag=fr, want_host=True)
        df.free()
        assert np.abs(fr.get_eri_s4() - out).max() == 0.0 and fr.mo_route_used() == (False, src.j2c.shape[0])
        rng = np.random.default_rng(5)
        A = rng.standard_normal((n, n))
        h = (np.diag(2.0 * np.arange(n)) + 0.15 * (A + A.T)) * 8.0 * max(1.0, float(np.abs(out).max()))      # (a gap the synthetic integrals cannot close)
        fr.set_energy_data(h, 0.1 * (A + A.T), None, 1.0, [0])
        o = max(1, n // 3)
        opts = default_opts(lib, cc_conv_tol=1e-12, cc_conv_tol_normt=1e-10)
        res = {}
        for route in (0, 1):
            fr.set_mo_route(route)
            res[route] = fr.solve(o, h, opts=opts, eeval=True)
            assert fr.mo_route_used()[0] == bool(route)
        assert abs(res[0]["e_corr_mo"] - res[1]["e_corr_mo"]) < 1e-10 and np.abs(res[0]["e_frag"] - res[1]["e_frag"]).max() < 1e-10
        assert np.abs(res[0]["rdm1_emb"] - res[1]["rdm1_emb"]).max() < 1e-9
        # the Cholesky factor handed in instead of (P|Q): the `build_lowtri_PQ` seam (eri_sparse_DF.py:535-556)
        df = et.DFContext(L_PQ=np.linalg.cholesky(src.j2c), lib=lib)
        df.set_ints(layouts["packed"], N, "packed")
        ref = G[f"{name}/eri0"]
        assert np.abs(df.transform(TAs[0], want_host=True) - ref).max() < tol * max(1.0, np.abs(ref).max())
        df.free()


def test_c_abi_on_mock_matches_reference_integral_direct_DF():
    import build as hc_build
    from quemb_amd import _lib
    check_df_golden(_lib.declare(C.CDLL(str(hc_build.build()))))
