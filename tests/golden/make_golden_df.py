"""Golden vectors for the molecular direct DF transform: tests/golden/df.npz.

RUNS the reference's own `quemb.molbe.eri_onthefly.integral_direct_DF` and `block_step_size` (molbe/eri_onthefly.py:18-42, :45-145) in the
build container.  PySCF is not installed, so the names that module imports from it are bound to providers of SYNTHETIC integral values
(tests/df_source.py, data only): `make_auxmol` / `auxmol.intor("int2c2e")` / `getints3c("int3c2e", ..., shls_slice)` return seeded
arrays with the symmetries of the real quantities, `mole.conc_env` / `make_loc` / `make_cintopt` return placeholders that are only
passed through, and `lib.prange` / `lib.map_with_prefetch` / `restore('4', ...)` are served by their plain definitions.  Everything
between those calls -- the block ranges from block_step_size and settings.INTEGRAL_TRANSFORM_MAX_MEMORY, the shell slices, the two
rotations into each fragment space, the placement of the blocks, the Cholesky factor, the triangular solve, bb^T bb -- is the
reference's code (scipy's cholesky / solve_triangular are the real ones).  Outputs are data only: the fragment ERIs (4-fold packed) and the
block ranges the reference chose.  Inputs are regenerated from the seeds in tests/df_source.py by the tests.

    python tests/golden/make_golden_df.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import make_golden as mg  # noqa: E402
from df_source import CASES, make_case  # noqa: E402


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def run_reference(eo, src, TAs, shells_per_block):
    from qemb_oracle import eri as oeri
    nshell_orb = 4                                                   # orbital shells of the "molecule" (only offsets the aux shell indices)
    nshell_aux = len(src.aux_shell_sizes)
    mol = _Obj(nbas=nshell_orb, nao=src.nao, _atm="atm", _bas="bas", _env="env", _add_suffix=lambda name: name)
    auxmol = _Obj(nbas=nshell_aux, nao=src.naux, _atm="aatm", _bas="abas", _env="aenv",
                  intor=lambda name, hermi=0: (src.j2c.copy() if name == "int2c2e" and hermi == 1 else None))
    seen = []

    def getints3c(intor, atm, bas, env, shls_slice, comp, aosym, ao_loc, cintopt, out=None):
        assert intor == "int3c2e" and (atm, bas, env) == ("ATM", "BAS", "ENV") and comp == 1 and aosym == "s1"
        assert ao_loc == "AO_LOC" and cintopt == "CINTOPT" and shls_slice[:4] == (0, nshell_orb, 0, nshell_orb)
        s0, s1 = shls_slice[4] - nshell_orb, shls_slice[5] - nshell_orb
        seen.append((s0, s1))
        return src.block(s0, s1)

    eo.make_auxmol = lambda m, auxbasis=None: auxmol
    eo.mole = _Obj(conc_env=lambda *a: ("ATM", "BAS", "ENV"))
    eo.make_loc = lambda bas, name: "AO_LOC"
    eo.make_cintopt = lambda atm, bas, env, name: "CINTOPT"
    eo.getints3c = getints3c
    eo.restore = lambda sym, e, n: oeri.pack_s4(np.asarray(e).reshape(n, n, n, n))
    eo.lib = _Obj(prange=lambda a, b, s: ((i, min(b, i + s)) for i in range(a, b, s)), map_with_prefetch=lambda f, it: map(f, it))
    # the reference's own block_step_size runs; the memory setting is chosen so that it yields `shells_per_block`
    nfrag = len(TAs)
    per_shell = 8.0 * src.nao * src.nao * nshell_aux * nfrag
    mem_gb = (per_shell * (shells_per_block + 0.5) if shells_per_block else per_shell * 1000) / 1e9
    eo.settings = _Obj(INTEGRAL_TRANSFORM_MAX_MEMORY=mem_gb)
    Fobjs = [_Obj(TA=TA, nao=TA.shape[1], dname=f"f{i}") for i, TA in enumerate(TAs)]
    store = {}
    eo.integral_direct_DF(_Obj(mol=mol), Fobjs, _Obj(create_dataset=lambda name, data=None: store.__setitem__(name, np.array(data))), auxbasis="synthetic")
    return [store[f"f{i}"] for i in range(len(TAs))], seen


def main():
    mg._install()
    import quemb.molbe.eri_onthefly as eo
    out = {}
    for name in CASES:
        src, TAs, step = make_case(name)
        eris, blocks = run_reference(eo, src, TAs, step)
        want = step or len(src.aux_shell_sizes)
        assert blocks == [(i, min(len(src.aux_shell_sizes), i + want)) for i in range(0, len(src.aux_shell_sizes), want)], blocks
        out[f"{name}/blocks"] = np.array(blocks)
        for i, e in enumerate(eris):
            out[f"{name}/eri{i}"] = e
        print(name, "blocks", blocks, [e.shape for e in eris])
    np.savez_compressed(HERE / "df.npz", **out)
    print("wrote", HERE / "df.npz")


if __name__ == "__main__":
    main()
