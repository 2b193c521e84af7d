"""CPU: the host integral source (quemb_amd/integrals.py + csrc_host/gto_ints.c) for shells beyond p -- d orbital shells and
auxiliary shells up to g as real solid harmonics -- against closed forms, numerical quadrature and invariances.  (The s/p part is
pinned by the reference's golden energies: octane E_HF, H8, tests/test_hostlogic_be.py.)"""
import numpy as np
import pytest
from scipy import integrate

from quemb_amd.integrals import Mole, aux_e2, aux_e2_pairs, cart2sph, cart_components, etb_auxbasis, int2c2e, make_auxmol


def test_cart2sph_spans_the_harmonic_polynomials():
    for l in range(5):
        c = cart2sph(l)
        comps = cart_components(l)
        assert c.shape == (len(comps), 2 * l + 1) and np.linalg.matrix_rank(c) == 2 * l + 1
        # evaluate on random points: every column is a harmonic function (numerical Laplacian == 0) of degree l (homogeneous)
        rng = np.random.default_rng(l)
        for col in c.T:
            f = lambda x: sum(w * x[0] ** a * x[1] ** b * x[2] ** d for w, (a, b, d) in zip(col, comps))
            x0 = rng.standard_normal(3)
            h = 1e-3
            lap = sum((f(x0 + h * e) - 2 * f(x0) + f(x0 - h * e)) / h ** 2 for e in np.eye(3))
            assert abs(lap) < 1e-5 * max(1.0, abs(f(x0)))
            assert abs(f(2.0 * x0) - 2.0 ** l * f(x0)) < 1e-9 * max(1.0, abs(f(x0)))


def test_single_centre_spdfg_functions_are_orthonormal():
    basis = {"C": [(l, [0.9, 0.31], [0.6, 0.5]) for l in range(5)] + [(2, [1.7], [1.0]), (4, [0.45], [1.0])]}
    mol = Mole([["C", (0.1, -0.2, 0.3)]], basis=basis)
    S = mol.one_electron()[0]
    assert mol.nao == 1 + 3 + 5 + 7 + 9 + 5 + 9
    # unit diagonal, and zero between different (l, m); two shells of the same l overlap only between the same m
    assert np.abs(np.diag(S) - 1.0).max() < 1e-12
    loc = mol.ao_loc_nr()
    for i in range(mol.nbas):
        for j in range(i):
            blk = S[loc[i]: loc[i + 1], loc[j]: loc[j + 1]]
            if mol.bas_angular(i) != mol.bas_angular(j):
                assert np.abs(blk).max() < 1e-12
            else:
                assert np.abs(blk - np.diag(np.diag(blk))).max() < 1e-12 and np.ptp(np.diag(blk)) < 1e-12


@pytest.mark.parametrize("l", [0, 1, 2, 3, 4])
def test_same_centre_coulomb_metric_against_radial_quadrature(l):
    """(P|Q) between two primitive solid-harmonic Gaussians r^l Y_lm exp(-a r^2) on one centre: by the Laplace expansion of 1/r12 only
    equal (l, m) couple, with (4 pi / (2l+1)) int int R_a(r1) R_b(r2) r_<^l / r_>^(l+1) r1^2 r2^2 dr1 dr2 -- a 2-D quadrature."""
    a, b = 0.8, 1.9
    aux = Mole([["H", (0.0, 0.0, 0.0)]], basis={"H": [(l, [a], [1.0]), (l, [b], [1.0])]})
    J = int2c2e(aux)
    n = 2 * l + 1
    blk = J[:n, n:]
    assert np.abs(blk - np.diag(np.diag(blk))).max() < 1e-12 and np.ptp(np.diag(blk)) < 1e-12
    # radial normalisation of r^l exp(-a r^2): int R^2 r^2 dr = 1
    nrm = lambda e: 1.0 / np.sqrt(integrate.quad(lambda r: r ** (2 * l + 2) * np.exp(-2 * e * r * r), 0, np.inf)[0])
    Ra = lambda r: nrm(a) * r ** l * np.exp(-a * r * r)
    Rb = lambda r: nrm(b) * r ** l * np.exp(-b * r * r)
    inner = lambda r1: (integrate.quad(lambda r2: Rb(r2) * r2 ** (l + 2), 0, r1)[0] / r1 ** (l + 1)
                        + r1 ** l * integrate.quad(lambda r2: Rb(r2) * r2 ** (1 - l), r1, np.inf)[0])
    ref = 4 * np.pi / (2 * l + 1) * integrate.quad(lambda r1: Ra(r1) * r1 ** 2 * inner(r1), 0, np.inf, epsabs=1e-12, epsrel=1e-11)[0]
    assert abs(blk[0, 0] - ref) < 1e-8 * abs(ref), (blk[0, 0], ref)


def _rot(seed):
    q = np.linalg.qr(np.random.default_rng(seed).standard_normal((3, 3)))[0]
    return q * np.sign(np.linalg.det(q))


def test_fitted_coulomb_energy_is_rotation_invariant_with_d_orbitals_and_g_auxiliaries():
    """E = sum D_pq (pq|P) (P|Q)^-1 (Q|rs) D_rs with D = S^-1 (a basis-independent operator) must not change when the molecule is
    rotated: mixes every Cartesian component of the d orbital shells and of the s..g auxiliary shells."""
    obs = {"C": [(0, [2.9, 0.68, 0.22], [0.2, 0.5, 0.4]), (1, [1.1, 0.3], [0.5, 0.6]), (2, [0.8], [1.0])],
           "H": [(0, [1.2, 0.3], [0.4, 0.7]), (1, [0.7], [1.0])]}
    geo = np.array([[0.0, 0.0, 0.0], [0.9, 0.3, -0.2], [-0.5, 0.8, 0.4]])
    vals = []
    for R in (np.eye(3), _rot(1), _rot(2)):
        mol = Mole([["C", tuple(R @ geo[0])], ["H", tuple(R @ geo[1])], ["H", tuple(R @ geo[2])]], basis=obs)
        aux = make_auxmol(mol, etb_auxbasis(mol, beta=2.5, lmax=4))
        assert max(aux.bas_angular(i) for i in range(aux.nbas)) == 4 and max(mol.bas_angular(i) for i in range(mol.nbas)) == 2
        S = mol.one_electron()[0]
        D = np.linalg.inv(S)
        j3 = aux_e2(mol, aux)
        j2 = int2c2e(aux)
        w = np.einsum("pqP,pq->P", j3, D)
        vals.append(float(w @ np.linalg.solve(j2, w)))
        assert np.linalg.eigvalsh(j2).min() > 0
    assert abs(vals[1] - vals[0]) < 1e-9 * abs(vals[0]) and abs(vals[2] - vals[0]) < 1e-9 * abs(vals[0]), vals


def test_pair_list_integrals_equal_the_dense_block_and_df_converges_to_the_exact_coulomb_energy():
    obs = {"C": [(0, [2.9, 0.68, 0.22], [0.2, 0.5, 0.4]), (1, [1.1, 0.3], [0.5, 0.6]), (2, [0.8], [1.0])],
           "H": [(0, [1.2, 0.3], [0.4, 0.7])]}
    mol = Mole([["C", (0.0, 0.0, 0.0)], ["H", (0.9, 0.3, -0.2)], ["H", (-0.5, 0.8, 0.4)]], basis=obs)
    aux = make_auxmol(mol, etb_auxbasis(mol, beta=2.0, lmax=4))
    j3 = aux_e2(mol, aux)
    pairs = [(p, q) for p in range(mol.nao) for q in range(p + 1) if (p + 2 * q) % 3]
    got = aux_e2_pairs(mol, aux, pairs)
    assert np.abs(got - np.array([j3[p, q] for p, q in pairs])).max() < 1e-12
    # exact Coulomb self-energy of the density D = S^-1 from the four-centre integrals vs the fitted one: the error falls with l_max
    eri = mol.eri_s1()
    D = np.linalg.inv(mol.one_electron()[0])
    exact = float(np.einsum("pq,pqrs,rs->", D, eri, D))
    errs = []
    for lmax in (1, 2, 4):
        a = make_auxmol(mol, etb_auxbasis(mol, beta=2.0, lmax=lmax))
        w = np.einsum("pqP,pq->P", aux_e2(mol, a), D)
        errs.append(exact - float(w @ np.linalg.solve(int2c2e(a), w)))
    assert all(e > -1e-10 for e in errs)                 # a Coulomb-metric fit never overshoots the self-energy
    assert errs[0] > errs[1] > errs[2] and errs[2] < 0.25 * errs[0], errs
