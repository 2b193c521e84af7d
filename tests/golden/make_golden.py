"""Generate tests/golden/*.npz from the REFERENCE's own pure-NumPy functions.

Run in the build container only (needs /root/reference; nothing here runs on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported normally (pyscf, h5py, numba, ... are not installed), so the missing
third-party roots are replaced by inert stub modules and `quemb/__init__.py` is bypassed with bare package
shells; only functions that are pure NumPy are then CALLED.  `h5py.File` reads inside get_frag_energy /
update_ebe_hf are served from an in-memory dict.  Outputs are data only (inputs + expected outputs).
"""
import importlib
import importlib.abc
import importlib.machinery
import json
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
MISSING = {"pyscf", "h5py", "numba", "chemcoord", "pathos", "libdmet", "cattrs", "ordered_set", "networkx", "attrs", "attr"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING or name.startswith("quemb.molbe._cpp"):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def _install():
    for r in list(MISSING):
        try:
            importlib.import_module(r)
            MISSING.discard(r)
        except Exception:
            pass
    sys.meta_path.insert(0, _StubFinder())
    for pk in ["quemb", "quemb.molbe", "quemb.kbe", "quemb.shared", "quemb.shared.external"]:
        m = types.ModuleType(pk)
        m.__path__ = [str(REF / "src" / pk.replace(".", "/"))]
        sys.modules[pk] = m
    import numba
    numba.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))


class _FakeH5File:
    """Serves `with h5py.File(name) as f: f[dname][()]` from a dict of arrays."""
    store = {}

    def __init__(self, name, mode="r"):
        self.name = str(name)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getitem__(self, key):
        arr = self.store[key]

        class _DS:
            def __getitem__(self, idx):
                return arr[idx] if idx != () else arr
        return _DS()

    def get(self, key):
        return self.store[key]


def frag_lists(data):
    """relAO_per_edge / relAO_in_ref_per_edge derived from the fixture lists (autofrag.py:554-698)."""
    AO = data["AO_per_frag"]
    rel_edge = [[[AO[i].index(a) for a in e] for e in data["AO_per_edge_per_frag"][i]] for i in range(len(AO))]
    rel_ref = [[[AO[r].index(a) for a in e] for e, r in zip(data["AO_per_edge_per_frag"][i], data["ref_frag_idx_per_edge_per_frag"][i])]
               for i in range(len(AO))]
    return rel_edge, rel_ref


def main():
    _install()
    sys.path.insert(0, str(REF / "tests"))
    from _expected_data_for_fragmentation_test import get_expected
    from quemb.kbe.solver import schmidt_decomp_svd
    from quemb.molbe import helper as ref_helper
    from quemb.molbe import pfrag as ref_pfrag
    from quemb.molbe.mbe import initialize_pot
    from quemb.molbe.pfrag import Frags, schmidt_decomposition
    from quemb.molbe.solver import solve_error
    from quemb.shared.external.ccsd_rdm import make_rdm1_ccsd_t1, make_rdm2_urlx
    from quemb.shared.external.optqn import FrankQN
    from quemb.shared.helper import ravel_symmetric, unravel_symmetric

    ref_helper.h5py = types.SimpleNamespace(File=_FakeH5File)
    ref_pfrag.h5py = types.SimpleNamespace(File=_FakeH5File)

    # ---- fragmentation fixtures (data only) -> json ---------------------------------------------
    frag_json = {}
    for key in ["test_autogen_h_linear_be1", "test_autogen_h_linear_be2", "test_autogen_h_linear_be3",
                "test_autogen_octane_be1", "test_autogen_octane_be2", "test_autogen_octane_be3"]:
        d = get_expected(key)
        rel_edge, rel_ref = frag_lists(d)
        frag_json[key] = dict(
            AO_per_frag=[list(map(int, x)) for x in d["AO_per_frag"]],
            AO_per_edge_per_frag=[[list(map(int, e)) for e in x] for x in d["AO_per_edge_per_frag"]],
            ref_frag_idx_per_edge_per_frag=[list(map(int, x)) for x in d["ref_frag_idx_per_edge_per_frag"]],
            relAO_per_origin_per_frag=[list(map(int, x)) for x in d["relAO_per_origin_per_frag"]],
            weight_and_relAO_per_center_per_frag=[[float(w), list(map(int, c))] for w, c in d["weight_and_relAO_per_center_per_frag"]],
            relAO_per_edge_per_frag=rel_edge, relAO_in_ref_per_edge_per_frag=rel_ref)
    frag_json["energies"] = {k: get_expected(k) for k in ["test_graphgen_autogen_h_linear_be2", "test_graphgen_autogen_octane_be2"]}
    (OUT / "fragmentation.json").write_text(json.dumps(frag_json))

    # ---- a1 / a1': Schmidt ---------------------------------------------------------------------------
    out = {}
    rng = np.random.default_rng(20260801)
    for case, (N, nocc, frag) in enumerate([(12, 5, [0, 1, 2]), (30, 11, [4, 3, 5, 9]), (58, 33, list(range(21))), (40, 10, [39, 0, 7])]):
        C = np.linalg.qr(rng.standard_normal((N, N)))[0]
        TA, nf, nb = schmidt_decomposition(C, nocc, frag, thr_bath=1e-10)
        D = C[:, :nocc] @ C[:, :nocc].T
        TAs = schmidt_decomp_svd(D, frag, thr_bath=1e-10)
        out[f"C{case}"] = C; out[f"nocc{case}"] = nocc; out[f"frag{case}"] = np.array(frag)
        out[f"TA{case}"] = TA; out[f"nfnb{case}"] = np.array([nf, nb]); out[f"TAsvd{case}"] = TAs.real
    np.savez_compressed(OUT / "schmidt.npz", **out)

    # ---- a9: RDMs --------------------------------------------------------------------------------------
    out = {}
    for case, (o, v) in enumerate([(2, 3), (4, 5), (3, 7)]):
        t1 = 0.1 * rng.standard_normal((o, v))
        t2 = 0.1 * rng.standard_normal((o, o, v, v)); t2 = t2 + t2.transpose(1, 0, 3, 2)
        out[f"t1_{case}"] = t1; out[f"t2_{case}"] = t2
        out[f"rdm1_{case}"] = make_rdm1_ccsd_t1(t1)
        out[f"rdm2_dm1_{case}"] = make_rdm2_urlx(t1, t2, with_dm1=True)
        out[f"rdm2_cum_{case}"] = make_rdm2_urlx(t1, t2, with_dm1=False)
    np.savez_compressed(OUT / "rdm.npz", **out)

    # ---- a12 / a13 / a11 / a15 on the H8 BE2 and octane BE2 fragment lists ------------------------------
    out = {}
    for tag, key in [("h8", "test_autogen_h_linear_be2"), ("oct", "test_autogen_octane_be2")]:
        d = get_expected(key)
        rel_edge, rel_ref = frag_lists(d)
        nfrag = len(d["AO_per_frag"])
        Fobjs = []
        for I in range(nfrag):
            f = Frags(d["AO_per_frag"][I], I, d["AO_per_edge_per_frag"][I], d["ref_frag_idx_per_edge_per_frag"][I],
                      rel_edge[I], rel_ref[I], d["weight_and_relAO_per_center_per_frag"][I], d["relAO_per_origin_per_frag"][I],
                      eri_file="fake.h5")
            Fobjs.append(f)
        pot = initialize_pot(nfrag, rel_edge)
        out[f"{tag}_npot"] = len(pot)
        u = rng.standard_normal(len(pot))
        out[f"{tag}_u"] = u
        cout = 0
        for I, f in enumerate(Fobjs):
            nf = len(d["AO_per_frag"][I])
            n = nf + max(1, nf - 1)          # some bath orbitals behind the fragment sites
            f.h1 = np.zeros((n, n))
            f.udim = cout
            cout = f.set_udim(cout)
            f.update_heff(u)
            out[f"{tag}_heff{I}"] = f.heff.copy()
            f.update_heff(u, only_chem=True)
            out[f"{tag}_heffchem{I}"] = f.heff.copy()
            R = rng.standard_normal((n, n)); f._rdm1 = 0.5 * (R + R.T)
            out[f"{tag}_rdm1_{I}"] = f._rdm1.copy()
            out[f"{tag}_udim{I}"] = f.udim
        Nocc = 4 if tag == "h8" else 33
        nrm, vec = solve_error(Fobjs, Nocc)
        out[f"{tag}_errnorm"] = nrm; out[f"{tag}_errvec"] = vec
        nrm, vec = solve_error(Fobjs, Nocc, only_chem=True)
        out[f"{tag}_errnorm_chem"] = nrm; out[f"{tag}_errvec_chem"] = vec
    # energies on one small fragment
    for case, (n, o, nf, cen) in enumerate([(6, 3, 3, [0, 1]), (9, 4, 4, [1])]):
        npr = n * (n + 1) // 2
        B = rng.standard_normal((2 * n, n, n)); B = B + B.transpose(0, 2, 1)
        eri1 = np.einsum("Ppq,Prs->pqrs", B, B) * 0.05
        il = np.tril_indices(n)
        eri4 = eri1[il[0], il[1]][:, il[0], il[1]]
        _FakeH5File.store = {"f0": eri4}
        mo = np.linalg.qr(rng.standard_normal((n, n)))[0]
        h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
        veff0 = rng.standard_normal((n, n)); veff0 = veff0 + veff0.T
        veff = rng.standard_normal((n, n)); veff = veff + veff.T
        v = n - o
        t1 = 0.1 * rng.standard_normal((o, v)); t2 = 0.1 * rng.standard_normal((o, o, v, v)); t2 = t2 + t2.transpose(1, 0, 3, 2)
        rdm1 = make_rdm1_ccsd_t1(t1)
        TA = np.zeros((n + 3, n))
        for cum in (True, False):
            rdm2 = make_rdm2_urlx(t1, t2, with_dm1=not cum)
            e = ref_helper.get_frag_energy(mo, o, nf, (1.0, cen), TA, h1, rdm1, rdm2, "f0", veff0, veff, cum, "fake.h5")
            out[f"efrag{case}_{int(cum)}"] = np.array(e)
        f = Frags(list(range(nf)), 0, [], [], [], [], (1.0, cen), [0], eri_file="fake.h5")
        f.h1 = h1; f.veff = veff; f.TA = TA; f._mo_coeffs = mo; f.nsocc = o
        e_h1, e_coul, e_vec = f.update_ebe_hf(return_e=True)
        out[f"ebehf{case}"] = np.array([f.ebe_hf, e_h1, e_coul]); out[f"ebehf_vec{case}"] = e_vec
        for k, a in dict(eri4=eri4, mo=mo, h1=h1, veff0=veff0, veff=veff, t1=t1, t2=t2).items():
            out[f"e{case}_{k}"] = a
        out[f"e{case}_meta"] = np.array([n, o, nf] + cen)
    np.savez_compressed(OUT / "be_pieces.npz", **out)

    # ---- index helpers + QN trajectory --------------------------------------------------------------------
    out = {}
    out["ravel"] = np.array([[ravel_symmetric(a, b) for b in range(12)] for a in range(12)])
    out["unravel"] = np.array([unravel_symmetric(i) for i in range(78)])
    A = rng.standard_normal((5, 5)) + 4 * np.eye(5)
    b = rng.standard_normal(5)

    def func(x):
        return A @ x + 0.1 * np.tanh(x) - b
    x0 = np.zeros(5)
    J0 = A + 0.3 * rng.standard_normal((5, 5))
    for tr in (False, True):
        qn = FrankQN(func, x0.copy(), func(x0), J0, max_space=20)
        xs = []
        for it in range(8):
            qn.next_step(it, trust_region=tr)
            xs.append(qn.xnew.copy())
        out[f"qn_xs_{int(tr)}"] = np.array(xs)
    out["qn_A"] = A; out["qn_b"] = b; out["qn_J0"] = J0
    np.savez_compressed(OUT / "misc.npz", **out)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
