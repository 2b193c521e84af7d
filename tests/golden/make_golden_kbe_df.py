"""Golden vectors for the Gamma-point periodic direct DF transform: tests/golden/kbe_df.npz.

RUNS the reference's own `quemb.kbe.eri_onthefly.integral_direct_DF` (kbe/eri_onthefly.py:48-241) and `_j2c_cholesky_or_eig`
(:19-45) in the build container.  PySCF-PBC is not installed, so the names the reference imports from it are bound to providers
of SYNTHETIC integral values (tests/kbe_df_source.py, data only): `ft_aopair`, `ft_ao`, `aux_e2`, `get_coulG`, `_CCGDFBuilder.get_2c2e`,
`cell.get_Gv_weights` return seeded arrays with the symmetries of the real quantities; `lib.prange` / `lib.map_with_prefetch` /
`ao2mo.addons.restore('4', ...)` are served by their plain definitions.  Everything between those calls -- the block loops, the
rotation into the fragment spaces, the (L|G)(G|ij) contraction, the Cholesky / eigenvalue fit, bb^T bb, the imaginary-part test --
is the reference's code.  Output: per case the fragment ERIs (4-fold packed), the branch taken for the metric, or the fact that the
reference raised.  Inputs are regenerated from the seeds in tests/kbe_df_source.py by the tests.

    python tests/golden/make_golden_kbe_df.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import make_golden as mg  # noqa: E402
from kbe_df_source import CASES, make_case  # noqa: E402


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def run_reference(keo, src, TAs, g_step, shell_step):
    from qemb_oracle import eri as oeri
    nao, naux, nG = src.nao, src.naux, src.n_planewaves
    rng = np.random.default_rng(999)
    kws = 0.37
    coul = rng.uniform(0.5, 2.0, nG)
    Gv = np.zeros((nG, 3)); Gv[:, 0] = np.arange(nG)                 # row index carried in the first component
    rs_chg = 0.1 * rng.standard_normal((nao * nao, naux))             # what aux_e2(cell, chgcell) "returns"
    rs_aux = src._rs.reshape(naux, -1).T + rs_chg                     # aux_e2(cell, auxcell): difference = the source's real-space block
    auxcell = _Obj(nao=naux, nbas=len(src.aux_shell_sizes), tag="aux")
    chgcell = _Obj(nao=naux, nbas=len(src.aux_shell_sizes), tag="chg")
    cell = _Obj(nbas=nao, nao=nao, _add_suffix=lambda name: name, get_Gv_weights=lambda mesh: (Gv, None, kws))
    mf = _Obj(cell=cell, kpts=np.zeros((1, 3)))

    def aux_e2(c, aux, intor, aosym, shls_slice=None):
        f0, f1 = src.aux_ao_loc[shls_slice[4]], src.aux_ao_loc[shls_slice[5]]
        return (rs_aux if aux.tag == "aux" else rs_chg)[:, f0:f1].copy()

    class Builder:
        mesh = [3, 3, 3]

        def __init__(self, *a):
            pass

        def build(self):
            return self

        def get_2c2e(self, kpts):
            return [src.j2c().copy()]

    idx = lambda G: np.asarray(G)[:, 0].astype(int)
    keo.make_auxcell = lambda c, ab: auxcell
    keo.make_modrho_basis = lambda c, ab: chgcell
    keo._CCGDFBuilder = Builder
    keo.get_coulG = lambda c, mesh=None: coul.copy()
    keo.ft_aopair = lambda c, G: src._pw[idx(G)] / (coul[idx(G)] * kws).reshape(-1, 1, 1).conj()
    keo.ft_ao = lambda c, G: src._ft[idx(G)].copy()
    keo.aux_e2 = aux_e2
    keo.restore = lambda sym, e, n: oeri.pack_s4(np.asarray(e).reshape(n, n, n, n))
    keo.lib = _Obj(prange=lambda a, b, s: ((i, min(b, i + s)) for i in range(a, b, s)), map_with_prefetch=lambda f, it: map(f, it))
    keo.block_step_size = lambda nfrag, n, nao_, datatype=float: (g_step if datatype is not float else shell_step)
    Fobjs = [_Obj(TA=TA, nao=TA.shape[1], dname=f"f{i}") for i, TA in enumerate(TAs)]
    store = {}
    keo.integral_direct_DF(mf, Fobjs, _Obj(create_dataset=lambda name, data=None: store.__setitem__(name, np.array(data))), auxbasis="synthetic")
    return [store[f"f{i}"] for i in range(len(TAs))]


def main():
    mg._install()
    import quemb.kbe.eri_onthefly as keo
    out = {}
    for name in CASES:
        src, TAs, raises = make_case(name)
        fit, ischol = keo._j2c_cholesky_or_eig(src.j2c().copy())
        out[f"{name}/ischol"] = np.array(int(ischol))
        out[f"{name}/fit"] = np.asarray(fit)
        try:
            eris = run_reference(keo, src, TAs, g_step=4, shell_step=2)
            raised = False
        except ValueError as e:
            assert "Imaginary part of ERI" in str(e), e
            raised = True
        assert raised == raises, (name, raised)
        out[f"{name}/raised"] = np.array(int(raised))
        if not raised:
            for i, e in enumerate(eris):
                out[f"{name}/eri{i}"] = e
        print(name, "ischol", ischol, "raised", raised)
    np.savez_compressed(HERE / "kbe_df.npz", **out)
    print("wrote", HERE / "kbe_df.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
