"""Semi-sparse density-fitting pipeline from geometry alone -- host mirror of molbe/eri_sparse_DF.py's driver
(`_run_sparse_df_driver`, :535-656; `transform_sparse_DF_integral_gpu`, :686-706) and of molbe/eri_onthefly.py's dense
`integral_direct_DF` (:45-145), over the device transform that already consumes the reference's `SemiSparseSym3DTensor`.

Stages and where they run:
  approx_S_abs            (:928-959)   primitive absolute overlaps by Gauss-Hermite quadrature ON THE DEVICE (`qemb_abs_overlap_prim`;
                                       numba on the host in the reference), contraction |c|^T s |c| and normalisation on the host;
  _get_AO_per_AO          (:224-240)   thresholding of S_abs (and of S_abs |TA| for the per-fragment variant: one device matmul);
  get_sparse_P_mu_nu      (:410-494)   (P|mu nu) for the reachable unique AO pairs from the integral source (libcint shell blocks in the
                                       reference, libqemb_gto here: host in both), straight into the unique-pair storage;
  (P|Q), Cholesky, transform_integral  on the device (`DFContext`: qemb_df_create / qemb_df_set_ints_semisparse / qemb_df_transform_screened).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import eri_transform as et
from .integrals import Mole, aux_e2, aux_e2_pairs, cart_components, int2c2e, make_auxmol


def _primitive_shells(mol: Mole):
    """mol.decontract_basis(aggregate=True) (eri_sparse_DF.py:878-880): one primitive shell per (shell, primitive) and the matrix of
    |contraction coefficient x primitive normalisation| from primitive Cartesian functions to contracted Cartesian functions."""
    ls, exps, xyz, cart0, rows = [], [], [], [], []
    nprim_cart = 0
    for (ia, l, e, _c, _ao0, c0) in mol.shells:
        nc = len(cart_components(l))
        for k in range(len(e)):
            ls.append(l); exps.append(e[k]); xyz.append(mol.atom[ia][1]); cart0.append(nprim_cart)
            for c in range(nc):
                rows.append((nprim_cart + c, c0 + c, abs(mol.bfs[c0 + c].co[k])))
            nprim_cart += nc
    A = np.zeros((nprim_cart, mol.ncart))
    for r, c, v in rows:
        A[r, c] = v
    return (np.array(ls, dtype=np.int32), np.array(exps), np.ascontiguousarray(xyz, dtype=float), np.array(cart0, dtype=np.int64), nprim_cart, A)


def approx_S_abs(mol: Mole, nroots: int = 500, lib=None) -> np.ndarray:
    """eri_sparse_DF.py:928-959: the approximated absolute overlap matrix int |phi_i| |phi_j| -- exact for uncontracted Cartesian
    functions, bounded by the triangle inequality through contractions and the Cartesian -> spherical transformation -- normalised to
    a unit diagonal (`_ensure_normalization`, :962-964; PySCF molecules are spherical by default, so the reference takes that branch)."""
    lib = lib or _lib.init()
    ls, exps, xyz, cart0, npc, A = _primitive_shells(mol)
    from scipy.special import roots_hermite                               # (:888; numpy's hermgauss overflows at 500 points)
    roots, weights = roots_hermite(int(nroots))
    roots = np.ascontiguousarray(roots); weights = np.ascontiguousarray(weights)
    s = np.empty((npc, npc))
    _lib.check(lib.qemb_abs_overlap_prim(len(ls), ls.ctypes.data, exps.ctypes.data, xyz.ctypes.data, cart0.ctypes.data, npc, int(nroots),
                                         roots.ctypes.data, weights.ctypes.data, s.ctypes.data), "qemb_abs_overlap_prim", lib)
    B = A @ np.abs(mol.c2s)                                               # |ctr_mat @ cart2spher| (the two factors do not overlap)
    S = B.T @ s @ B
    N = np.sqrt(np.diag(S))
    return S / (N[:, None] * N[None, :])


def _get_AO_per_AO(S_abs, epsilon: float, TA=None, lib=None) -> dict[int, list[int]]:
    """eri_sparse_DF.py:224-240: AOs nu with S_abs[nu, mu] >= epsilon for every source AO mu; with TA only the AOs that the fragment's
    embedding orbitals reach, (S_abs |TA|).max(axis=1) > epsilon, are sources."""
    S_abs = np.asarray(S_abs)
    if TA is None:
        sources = range(len(S_abs))
    else:
        X = et.matmul(S_abs, np.abs(np.asarray(TA)), lib=lib)
        sources = np.nonzero(X.max(axis=1) > epsilon)[0]
    return {int(i): [int(x) for x in np.nonzero(S_abs[:, i] >= epsilon)[0]] for i in sources}


def get_sparse_P_mu_nu(mol: Mole, auxmol: Mole, exch_reachable) -> et.SemiSparseSym3DTensor:
    """eri_sparse_DF.py:410-494: the 3-centre integrals of the reachable AO pairs in the reference's semi-sparse storage."""
    reach = [sorted(exch_reachable.get(mu, [])) for mu in range(mol.nao)]
    # the storage needs a symmetric relation (indexers.hpp:149-162); S_abs is symmetric, the per-fragment source selection is not
    sym = [set(r) for r in reach]
    for mu, r in enumerate(reach):
        for nu in r:
            sym[nu].add(mu)
    t = et.SemiSparseSym3DTensor((auxmol.nao, mol.nao, mol.nao), [sorted(x) for x in sym])
    pairs = sorted(((mu, nu) for mu, r in enumerate(t.exch_reachable_unique) for nu in r), key=lambda pq: t.offsets[et.ravel_symmetric(*pq)])
    if pairs:
        vals = aux_e2_pairs(mol, auxmol, pairs)                        # (n_unique, naux), in offset order
        t.unique_dense_data[:, :] = vals.T
    assert not np.isnan(t.unique_dense_data).any()
    return t


def transform_sparse_DF_integral_hip(mf, Fobjs, auxbasis, AO_coeff_epsilon: float = 1e-10, MO_coeff_epsilon: float = 1e-5,
                                     precompute_P_mu_nu: bool = True, lib=None, stats=None, factor_only: bool = False):
    """`_run_sparse_df_driver` (eri_sparse_DF.py:535-656) with the device transform injected, as `transform_sparse_DF_integral_gpu`
    (:686-706) does with its cuBLAS one.  The fragment ERIs go straight into each fragment's device handle (the reference writes dataset
    `f{I}` of eri_file.h5).  Defaults as BE.__init__ (mbe.py:188-189)."""
    mol = mf.mol
    auxmol = make_auxmol(mol, auxbasis)
    S_abs = approx_S_abs(mol, lib=lib)
    df = et.DFContext(j2c=int2c2e(auxmol), lib=lib)                      # (P|Q) -> Cholesky on the device (build_lowtri_PQ)
    try:
        if precompute_P_mu_nu:
            P_mu_nu = get_sparse_P_mu_nu(mol, auxmol, _get_AO_per_AO(S_abs, AO_coeff_epsilon, None, lib=lib))
            df.set_ints_semisparse(P_mu_nu)
            if stats is not None:
                stats.update(n_unique=P_mu_nu.unique_dense_data.shape[1], n_pairs_dense=mol.nao * (mol.nao + 1) // 2, naux=auxmol.nao)
        for f in Fobjs:
            if not precompute_P_mu_nu:                                   # "on-fly-sparse-DF": only what this fragment reaches
                P_mu_nu = get_sparse_P_mu_nu(mol, auxmol, _get_AO_per_AO(S_abs, AO_coeff_epsilon, f.TA, lib=lib))
                df.set_ints_semisparse(P_mu_nu)
            df.transform(f.TA, frag=f.dev, want_host=False, S_abs=S_abs, MO_coeff_epsilon=MO_coeff_epsilon, factor_only=factor_only)
    finally:
        df.free()
    return S_abs


def integral_direct_DF_hip(mf, Fobjs, auxbasis, lib=None, factor_only: bool = False):
    """molbe/eri_onthefly.py:45-145 (`int-direct-DF`): dense (mu nu|P), fragment transform, fit with the Cholesky factor of (P|Q).  No
    auxiliary-index blocking: the blocks of :18-42 exist to stay inside host RAM; naux N^2 doubles fit in HBM."""
    mol = mf.mol
    auxmol = make_auxmol(mol, auxbasis)
    df = et.DFContext(j2c=int2c2e(auxmol), lib=lib)
    try:
        df.set_ints(aux_e2(mol, auxmol), mol.nao, "pqL")
        for f in Fobjs:
            df.transform(f.TA, frag=f.dev, want_host=False, factor_only=factor_only)
    finally:
        df.free()
