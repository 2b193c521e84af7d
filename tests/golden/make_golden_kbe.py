"""Golden vectors for the periodic front-end: the reference's OWN kbe.pfrag.Frags.sd / cons_h1 / get_nsocc
and kbe.misc.get_phase / get_phase1 run on a synthetic 1-D periodic tight-binding model.  Build container only:

    python tests/golden/make_golden_kbe.py        -> tests/golden/kbe.npz   (inputs + expected outputs, data only)

Same import stubs as make_golden.py; `pyscf.lib.cartesian_prod` (third party, absent) is served by an equivalent
numpy implementation, and `cell` is an object exposing lattice_vectors() only.
"""
import itertools
import sys
import types
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg

OUT = Path(__file__).resolve().parent


def cartesian_prod(arrays):
    return np.array(list(itertools.product(*[np.asarray(a) for a in arrays])))


def model(nk, nlo, nocc, seed):
    """Real symmetric block-circulant H and S on a ring of nk cells -> k-space lao (Loewdin), lmo, C, h1_k, S_k."""
    rng = np.random.default_rng(seed)
    a = 2.5
    a_vec = np.diag([a, 12.0, 12.0])
    kmesh = [nk, 1, 1]
    kpts = np.array([[2 * np.pi * m / (nk * a), 0.0, 0.0] for m in range(nk)])
    # h(R), s(R) for R = 0, +-1 (nearest cells); h(-R) = h(R)^T
    h0 = rng.standard_normal((nlo, nlo)); h0 = 0.5 * (h0 + h0.T) + np.diag(2.0 * np.arange(nlo))
    h1 = 0.3 * rng.standard_normal((nlo, nlo))
    s1 = 0.05 * rng.standard_normal((nlo, nlo))
    Hk, Sk = [], []
    for k in kpts:
        ph = np.exp(1j * k[0] * a)
        Hk.append(h0 + h1 * ph + h1.T * np.conj(ph))
        Sk.append(np.eye(nlo) + s1 * ph + s1.T * np.conj(ph))
    lao, lmo, C = [], [], []
    for H, S in zip(Hk, Sk):
        w, U = np.linalg.eigh(S)
        X = U @ np.diag(w ** -0.5) @ U.conj().T            # Loewdin orthogonalised AOs: lao^H S lao = 1
        e, V = np.linalg.eigh(X.conj().T @ H @ X)
        lao.append(X); lmo.append(V); C.append(X @ V)
    return dict(a_vec=a_vec, kmesh=np.array(kmesh), kpts=kpts, lao=np.array(lao), lmo=np.array(lmo), C=np.array(C),
                h1=np.array(Hk), S=np.array(Sk), nocc=nocc)


def main():
    mg._install()
    import quemb.kbe.misc as misc
    misc.cartesian_prod = cartesian_prod
    import quemb.kbe.pfrag as kp
    kp.get_phase, kp.get_phase1 = misc.get_phase, misc.get_phase1
    out = {}
    for case, (nk, nlo, nocc, frag, cen) in enumerate([(4, 3, 1, [0, 1], [0]), (3, 5, 2, [1, 2, 4], [1, 2]), (6, 4, 2, [0, 3], [0])]):
        m = model(nk, nlo, nocc, 100 + case)
        cell = types.SimpleNamespace(lattice_vectors=lambda a=m["a_vec"]: a)
        f = object.__new__(kp.Frags)
        f.AO_in_frag = frag; f.n_frag = len(frag); f.weight_and_relAO_per_center = (1.0, cen)
        f.sd(m["lao"], m["lmo"], nocc, 1e-10, cell=cell, kpts=m["kpts"], kmesh=list(m["kmesh"]), h1=m["h1"])
        f.cons_h1(m["h1"])
        P = f.get_nsocc(m["S"], m["C"], nocc)
        for k, v in m.items():
            out[f"c{case}_{k}"] = np.asarray(v)
        out[f"c{case}_frag"] = np.array(frag)
        out[f"c{case}_phase"] = misc.get_phase(cell, m["kpts"], list(m["kmesh"]))
        out[f"c{case}_phase1"] = misc.get_phase1(cell, m["kpts"], list(m["kmesh"]))
        out[f"c{case}_TA_lo_eo"] = f.TA_lo_eo; out[f"c{case}_TA"] = f.TA; out[f"c{case}_nao"] = f.nao
        out[f"c{case}_h1_eo"] = f.h1; out[f"c{case}_P"] = P; out[f"c{case}_nsocc"] = f.nsocc
        out[f"c{case}_mo_guess"] = f._mo_coeffs
    np.savez_compressed(OUT / "kbe.npz", **out)
    print("kbe golden vectors written:", {k: np.asarray(v).shape for k, v in out.items() if k.startswith("c0_")})


if __name__ == "__main__":
    main()
