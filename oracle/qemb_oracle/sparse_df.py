"""Oracle for the AO screening of the semi-sparse DF pipeline (test infrastructure).

Restates molbe/eri_sparse_DF.py:733-812 `_primitive_overlap` (absolute overlap of two uncontracted Cartesian shells by Gauss-Hermite
quadrature per direction) for UNNORMALISED primitives x^i y^j z^k exp(-a r^2); a brute-force 3-D grid integral pins it in
tests/test_oracle_sparse_df.py.  The reference's norm factors (:756-760, PySCF's Cartesian conventions) are left to the caller: the
product only enters `approx_S_abs` through |contraction coefficients| and is normalised to a unit diagonal at the end (:962-964)."""
import numpy as np


def cart_components(l):
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def primitive_abs_overlap(li, lj, ai, aj, Ra, Rb, roots, weights):
    """:763-812 without the normalisation prefactors."""
    Ra, Rb = np.asarray(Ra, float), np.asarray(Rb, float)
    aij = ai + aj
    Rp = (ai * Ra + aj * Rb) / aij
    scale = 1.0 / np.sqrt(aij)
    fac = scale ** 3 * np.exp(-ai * aj / aij * ((Ra - Rb) @ (Ra - Rb)))
    x = roots * scale + Rp[:, None]
    xa, xb = x - Ra[:, None], x - Rb[:, None]
    I = np.empty((3, li + 1, lj + 1))
    for d in range(3):
        for p in range(li + 1):
            for q in range(lj + 1):
                I[d, p, q] = np.sum(weights * np.abs(xa[d] ** p * xb[d] ** q))
    s = np.empty((len(cart_components(li)), len(cart_components(lj))))
    for i, (ix, iy, iz) in enumerate(cart_components(li)):
        for j, (jx, jy, jz) in enumerate(cart_components(lj)):
            s[i, j] = I[0, ix, jx] * I[1, iy, jy] * I[2, iz, jz] * fac
    return s


def abs_overlap_grid(li, lj, ai, aj, Ra, Rb, n=161, box=7.0):
    """the same integrals on a plain product grid (trapezoid rule): slow, independent of the quadrature above"""
    g = np.linspace(-box, box, n)
    w = np.full(n, g[1] - g[0]); w[0] = w[-1] = 0.5 * (g[1] - g[0])
    Ra, Rb = np.asarray(Ra, float), np.asarray(Rb, float)
    one = []
    for d in range(3):
        xa, xb = g - Ra[d], g - Rb[d]
        e = np.exp(-ai * xa * xa - aj * xb * xb)
        one.append(np.array([[np.sum(w * np.abs(xa ** p * xb ** q) * e) for q in range(lj + 1)] for p in range(li + 1)]))
    s = np.empty((len(cart_components(li)), len(cart_components(lj))))
    for i, (ix, iy, iz) in enumerate(cart_components(li)):
        for j, (jx, jy, jz) in enumerate(cart_components(lj)):
            s[i, j] = one[0][ix, jx] * one[1][iy, jy] * one[2][iz, jz]
    return s


def approx_S_abs(prim_shells, contraction, nroots=500):
    """:928-959 for primitive shells [(l, exponent, centre)] and the |coefficient| matrix (primitive Cartesian functions x AOs)."""
    from scipy.special import roots_hermite
    roots, weights = roots_hermite(nroots)
    off = np.cumsum([0] + [len(cart_components(l)) for l, _, _ in prim_shells])
    s = np.zeros((off[-1], off[-1]))
    for i, (li, ai, Ri) in enumerate(prim_shells):
        for j, (lj, aj, Rj) in enumerate(prim_shells[: i + 1]):
            b = primitive_abs_overlap(li, lj, ai, aj, Ri, Rj, roots, weights)
            s[off[i]: off[i + 1], off[j]: off[j + 1]] = b
            s[off[j]: off[j + 1], off[i]: off[i + 1]] = b.T
    S = np.abs(contraction).T @ s @ np.abs(contraction)
    N = np.sqrt(np.diag(S))
    return S / (N[:, None] * N[None, :])


def get_AO_per_AO(S_abs, epsilon, TA=None):
    """:224-240."""
    if TA is None:
        sources = range(len(S_abs))
    else:
        sources = ((S_abs @ np.abs(TA)).max(axis=1) > epsilon).nonzero()[0]
    return {int(i): [int(x) for x in (S_abs[:, i] >= epsilon).nonzero()[0]] for i in sources}
