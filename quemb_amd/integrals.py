"""Integral source + mean-field for real molecules without PySCF.

`Mole` / `RHF` expose the handful of PySCF attributes `BE` reads (molbe/mbe.py:361-373): `mol.nelectron`,
`mo_coeff`, `mo_energy`, `e_tot`, `_eri`, `energy_nuc()`, `get_hcore()`, `get_ovlp()`, `make_rdm1()`,
`get_veff()`, plus the shell queries the semi-sparse DF pipeline uses (molbe/eri_sparse_DF.py: `nbas`, `ao_loc_nr`, `bas_angular`,
`bas_exp`, `bas_coord`).  Integrals come from the in-tree host library libqemb_gto.so (csrc_host/gto_ints.c, McMurchie-Davidson over
Cartesian Gaussians: s, p, d orbital shells, auxiliary shells up to g); shells with l >= 2 are used as real solid harmonics like
PySCF's default (`cart=False`): the Cartesian integrals are contracted with a per-shell Cartesian -> spherical matrix.
Built-in orbital bases: STO-3G for H and C (the reference's test systems), cc-pVDZ for H; any basis can be passed as a dict
{symbol: [(l, exponents, coefficients), ...]}.  `etb_auxbasis` generates an even-tempered auxiliary basis -- the image holds no basis-set
library, so the reference's `weigend` (def2-universal-jfit) fitting basis is not available.
This is upstream of the hot path -- CPU work in the reference too (libcint) -- and exists so that the reference's
end-to-end golden energies can be reproduced from first principles.
"""

from __future__ import annotations

import ctypes as C
import itertools
from pathlib import Path

import numpy as np

BOHR = 0.52917721092            # Angstrom per Bohr (pyscf.data.nist.BOHR)
_HERE = Path(__file__).resolve().parent
GTO_LIB = _HERE / "libqemb_gto.so"
MAXPRIM = 8
_L = {"s": 0, "p": 1, "d": 2, "f": 3, "g": 4}

# STO-3G (EMSL / PySCF 'sto-3g'): (l, exponents, coefficients) per shell; 'sp' shells share exponents
_STO3G = {
    "H": [("s", [3.42525091, 0.62391373, 0.16885540], [0.15432897, 0.53532814, 0.44463454])],
    "C": [("s", [71.6168370, 13.0450960, 3.5305122], [0.15432897, 0.53532814, 0.44463454]),
          ("s", [2.9412494, 0.6834831, 0.2222899], [-0.09996723, 0.39951283, 0.70011547]),
          ("p", [2.9412494, 0.6834831, 0.2222899], [0.15591627, 0.60768372, 0.39195739])],
}
# cc-pVDZ, hydrogen only (2s1p; EMSL / PySCF 'cc-pvdz'): a multi-AO-per-atom test basis for the s/p generator
_CCPVDZ = {
    "H": [("s", [13.0100000, 1.9620000, 0.4446000], [0.0196850, 0.1379770, 0.4781480]),
          ("s", [0.1220000], [1.0]),
          ("p", [0.7270000], [1.0])],
}
_BASES = {"sto-3g": _STO3G, "cc-pvdz": _CCPVDZ}
_Z = {"H": 1, "C": 6}


class _BF(C.Structure):
    _fields_ = [("ctr", C.c_double * 3), ("lmn", C.c_int * 3), ("nprim", C.c_int), ("ex", C.c_double * MAXPRIM),
                ("co", C.c_double * MAXPRIM)]


def _dfact(n):
    return 1.0 if n <= 0 else float(np.prod(np.arange(n, 0, -2)))


def _load():
    if not GTO_LIB.exists():
        raise RuntimeError(f"{GTO_LIB} not found: build with __graft_entry__.build()")
    lib = C.CDLL(str(GTO_LIB))
    lib.gto_bf_size.restype = C.c_size_t
    assert lib.gto_bf_size() == C.sizeof(_BF)
    return lib


def read_xyz(path):
    lines = Path(path).read_text().strip().splitlines()
    n = int(lines[0])
    atoms = []
    for ln in lines[2: 2 + n]:
        s = ln.split()
        atoms.append((s[0], tuple(float(x) for x in s[1:4])))
    return atoms


def cart_components(l):
    """Cartesian monomial exponents of a shell in PySCF / libcint order: xx xy xz yy yz zz for l = 2, ..."""
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def cart2sph(l):
    """(ncart, 2l+1): real solid harmonics of degree l as combinations of the shell's Cartesian functions, for Cartesian functions
    that share ONE radial normalisation (the x^l component has unit norm).  The 2l+1 harmonic polynomials are the null space of the
    Laplacian on the degree-l monomials; the columns are orthonormalised under the Gaussian overlap metric of those monomials,
    <x^a y^b z^c | x^d y^e z^f> = (a+d-1)!! (b+e-1)!! (c+f-1)!! / (2l-1)!! for all-even sums.  Any orthonormal basis of that space gives
    the same fitted integrals (the density-fitting sums are invariant under rotations inside a shell); s and p are the identity."""
    comps = cart_components(l)
    nc = len(comps)
    if l <= 1:
        return np.eye(nc)
    lower = {c: i for i, c in enumerate(cart_components(l - 2))}
    Lap = np.zeros((len(lower), nc))
    for j, c in enumerate(comps):
        for d in range(3):
            if c[d] >= 2:
                cc = list(c); cc[d] -= 2
                Lap[lower[tuple(cc)], j] += c[d] * (c[d] - 1)
    _, sv, Vt = np.linalg.svd(Lap)
    null = Vt[len(lower):].T if len(lower) < nc else Vt[(sv > 1e-12).sum():].T           # nc x (2l+1)
    assert null.shape[1] == 2 * l + 1
    G = np.zeros((nc, nc))
    for i, a in enumerate(comps):
        for j, b in enumerate(comps):
            if all((a[d] + b[d]) % 2 == 0 for d in range(3)):
                G[i, j] = np.prod([_dfact(a[d] + b[d] - 1) for d in range(3)]) / _dfact(2 * l - 1)
    # Loewdin orthonormalisation under G: columns X with X^T G X = 1
    M = null.T @ G @ null
    w, U = np.linalg.eigh(M)
    return null @ (U / np.sqrt(w)) @ U.T


class Mole:
    def __init__(self, atom, basis="sto-3g", unit="Angstrom"):
        if isinstance(basis, str):
            if basis.lower() not in _BASES:
                raise NotImplementedError("built-in bases: STO-3G (H, C), cc-pVDZ (H); pass a dict {symbol: [(l, exps, coefs), ...]} otherwise")
            table = _BASES[basis.lower()]
        else:
            table = basis
        self.basis = basis
        if isinstance(atom, (str, Path)):
            atom = read_xyz(atom)
        scale = 1.0 / BOHR if unit.lower().startswith("a") else 1.0
        self.atom = [(sym, tuple(scale * np.asarray(xyz, dtype=float))) for sym, xyz in atom]
        self.nelectron = sum(_Z[s] for s, _ in self.atom)
        self.bfs = []          # Cartesian contracted functions handed to libqemb_gto
        self.ao_atom = []      # atom of every (spherical) AO
        self.shells = []       # (atom, l, exponents, coefficients, first AO, first Cartesian function)
        blocks = []
        for ia, (sym, xyz) in enumerate(self.atom):
            if sym not in table:
                raise NotImplementedError(f"no basis for {sym} in the table given")
            for l, exps, coefs in table[sym]:
                l = _L[l] if isinstance(l, str) else int(l)
                if l > 4:
                    raise NotImplementedError("shells beyond g are not supported")
                self.shells.append((ia, l, np.asarray(exps, dtype=float), np.asarray(coefs, dtype=float), len(self.ao_atom), len(self.bfs)))
                for lmn in cart_components(l):
                    self.bfs.append(self._make_bf(xyz, lmn, exps, coefs, common=(l >= 2)))
                blocks.append(cart2sph(l))
                self.ao_atom += [ia] * (2 * l + 1)
        self.ncart = len(self.bfs)
        self.nao = len(self.ao_atom)
        self.cart = all(b.shape[0] == b.shape[1] for b in blocks)      # only s and p shells: Cartesian == spherical
        self.c2s = np.zeros((self.ncart, self.nao))
        r = c = 0
        for b in blocks:
            self.c2s[r: r + b.shape[0], c: c + b.shape[1]] = b
            r += b.shape[0]; c += b.shape[1]

    @staticmethod
    def _make_bf(xyz, lmn, exps, coefs, common=False):
        """One Cartesian contracted Gaussian.  common=False (s, p): unit self-overlap.  common=True (l >= 2): every component of the
        shell carries the normalisation of its x^l component, the convention `cart2sph` assumes."""
        L = sum(lmn)
        nl = (L, 0, 0) if common else lmn
        e = np.asarray(exps, dtype=float); c = np.asarray(coefs, dtype=float)
        norm = (2 * e / np.pi) ** 0.75 * (4 * e) ** (L / 2.0) / np.sqrt(_dfact(2 * nl[0] - 1) * _dfact(2 * nl[1] - 1) * _dfact(2 * nl[2] - 1))
        cc = c * norm
        # renormalise the contraction to unit self-overlap
        pref = np.pi ** 1.5 * _dfact(2 * nl[0] - 1) * _dfact(2 * nl[1] - 1) * _dfact(2 * nl[2] - 1) / 2.0 ** L
        s = sum(cc[i] * cc[j] * pref / (e[i] + e[j]) ** (L + 1.5) for i in range(len(e)) for j in range(len(e)))
        cc = cc / np.sqrt(s)
        if len(e) > MAXPRIM:
            raise NotImplementedError(f"at most {MAXPRIM} primitives per contraction")
        b = _BF()
        b.ctr[:] = xyz; b.lmn[:] = lmn; b.nprim = len(e)
        for i in range(len(e)):
            b.ex[i] = e[i]; b.co[i] = cc[i]
        return b

    def _arr(self):
        return (_BF * self.ncart)(*self.bfs)

    # ---- shell queries (the PySCF names the reference's eri_sparse_DF.py uses) -----------------------------------------------
    @property
    def natm(self):
        return len(self.atom)

    @property
    def nbas(self):
        return len(self.shells)

    def ao_loc_nr(self):
        return np.array([sh[4] for sh in self.shells] + [self.nao])

    def bas_angular(self, i):
        return self.shells[i][1]

    def bas_exp(self, i):
        return self.shells[i][2]

    def bas_ctr_coeff(self, i):
        return self.shells[i][3]

    def bas_atom(self, i):
        return self.shells[i][0]

    def bas_coord(self, i):
        return np.asarray(self.atom[self.shells[i][0]][1])

    def atom_charge(self, ia):
        return _Z[self.atom[ia][0]]

    def aoslice_by_atom(self):
        """PySCF's (shell0, shell1, ao0, ao1) per atom."""
        ao = np.asarray(self.ao_atom)
        sh = np.asarray([s[0] for s in self.shells])
        out = []
        for ia in range(self.natm):
            w = np.nonzero(ao == ia)[0]
            ws = np.nonzero(sh == ia)[0]
            out.append((int(ws[0]), int(ws[-1]) + 1, int(w[0]), int(w[-1]) + 1))
        return np.array(out)

    def energy_nuc(self):
        e = 0.0
        for i, (si, xi) in enumerate(self.atom):
            for sj, xj in self.atom[:i]:
                e += _Z[si] * _Z[sj] / np.linalg.norm(np.asarray(xi) - np.asarray(xj))
        return e

    # ---- integrals ---------------------------------------------------------------------------------------------------------
    def _sph2(self, X):
        return X if self.cart else self.c2s.T @ X @ self.c2s

    def one_electron(self):
        lib = _load()
        n = self.ncart
        S = np.zeros((n, n)); T = np.zeros((n, n)); V = np.zeros((n, n))
        xyz = np.ascontiguousarray([a[1] for a in self.atom], dtype=float)
        Z = np.ascontiguousarray([_Z[a[0]] for a in self.atom], dtype=float)
        lib.gto_one_electron(n, self._arr(), len(self.atom), xyz.ctypes.data_as(C.c_void_p), Z.ctypes.data_as(C.c_void_p),
                             S.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
        return self._sph2(S), self._sph2(T), self._sph2(V)

    def eri_s1(self):
        lib = _load()
        n = self.ncart
        out = np.zeros((n, n, n, n))
        lib.gto_eri_s1(n, self._arr(), out.ctypes.data_as(C.c_void_p))
        if not self.cart:
            c = self.c2s
            out = np.einsum("pqrs,pi,qj,rk,sl->ijkl", out, c, c, c, c, optimize=True)
        return out

    def intor(self, name):
        if name == "int1e_ovlp":
            return self.one_electron()[0]
        if name == "int2c2e":
            return int2c2e(self)
        raise NotImplementedError(name)


# ---- auxiliary basis and the 2- / 3-centre Coulomb integrals of density fitting ---------------------------------------------
def etb_auxbasis(mol: Mole, beta=2.0, lmax=None, lmax_by_symbol=None):
    """An even-tempered auxiliary basis in the spirit of PySCF's `df.addons.aug_etb`: for every element and every auxiliary angular
    momentum L up to twice the largest orbital l (capped at `lmax`), uncontracted exponents emin * beta^k covering the range of the
    orbital-PRODUCT exponents a_i + a_j of the shell pairs (l_i, l_j) that can couple to L.  Returned as a basis dict for `Mole`."""
    out = {}
    for sym in sorted(set(s for s, _ in mol.atom)):
        ia = next(i for i, (s, _) in enumerate(mol.atom) if s == sym)
        shells = [(sh[1], sh[2]) for sh in mol.shells if sh[0] == ia]
        lo = max(l for l, _ in shells)
        Lmax = 2 * lo if lmax is None else min(2 * lo, lmax)
        if lmax_by_symbol and sym in lmax_by_symbol:
            Lmax = lmax_by_symbol[sym]
        basis = []
        for L in range(Lmax + 1):
            sums = [a + b for (li, ei), (lj, ej) in itertools.product(shells, shells) if abs(li - lj) <= L <= li + lj for a in ei for b in ej]
            if not sums:       # no orbital pair of this element reaches L (e.g. polarisation functions for a minimal basis): reuse L - 1
                sums = prev
            emin, emax = min(sums), max(sums)
            nexp = max(1, int(np.ceil(np.log(emax / emin) / np.log(beta))) + 1)
            basis += [(L, [emin * beta ** k], [1.0]) for k in range(nexp)]
            prev = sums
        out[sym] = basis
    return out


def make_auxmol(mol: Mole, auxbasis):
    """pyscf.df.addons.make_auxmol: the same atoms carrying the auxiliary basis (a dict, or "etb" / ("etb", beta, lmax))."""
    if isinstance(auxbasis, str):
        if auxbasis.lower() != "etb":
            raise NotImplementedError(f"auxiliary basis {auxbasis!r}: the image holds no basis-set library (no PySCF, no network); "
                                      "pass a basis dict or 'etb' (even-tempered, generated from the orbital basis)")
        auxbasis = etb_auxbasis(mol)
    elif isinstance(auxbasis, tuple) and auxbasis[0] == "etb":
        auxbasis = etb_auxbasis(mol, *auxbasis[1:])
    aux = Mole(mol.atom, basis=auxbasis, unit="Bohr")
    return aux


def int2c2e(auxmol: Mole):
    """(P|Q), auxmol.intor('int2c2e') (molbe/eri_onthefly.py:106-108, eri_sparse_DF.py:611)."""
    lib = _load()
    n = auxmol.ncart
    out = np.zeros((n, n))
    lib.gto_eri_2c(n, auxmol._arr(), out.ctypes.data_as(C.c_void_p))
    return auxmol._sph2(out)


def aux_e2(mol: Mole, auxmol: Mole):
    """(mu nu|P), dense (N, N, naux): pyscf.df.incore.aux_e2(mol, auxmol, 'int3c2e') (eri_onthefly.py:64-98)."""
    lib = _load()
    out = np.zeros((mol.ncart, mol.ncart, auxmol.ncart))
    lib.gto_eri_3c(mol.ncart, mol._arr(), auxmol.ncart, auxmol._arr(), out.ctypes.data_as(C.c_void_p))
    if not auxmol.cart:
        out = out @ auxmol.c2s
    if not mol.cart:
        out = np.einsum("pqP,pi,qj->ijP", out, mol.c2s, mol.c2s, optimize=True)
    return out


def aux_e2_pairs(mol: Mole, auxmol: Mole, pairs):
    """(mu nu|P) for a list of AO pairs only: (npairs, naux), one auxiliary vector per pair -- the fill of the semi-sparse tensor
    (get_sparse_P_mu_nu, eri_sparse_DF.py:410-494, which asks libcint for the shell blocks that contain the reachable pairs)."""
    lib = _load()
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    if mol.cart:
        cp = pairs
        W = None
    else:
        # a spherical AO is a combination of the Cartesian functions of its shell: compute every Cartesian pair of the shell pairs touched
        sh_of = np.repeat(np.arange(mol.nbas), [2 * s[1] + 1 for s in mol.shells])
        cart_of_shell = [np.arange(s[5], s[5] + len(cart_components(s[1]))) for s in mol.shells]
        need = sorted(set((int(sh_of[p]), int(sh_of[q])) for p, q in pairs))
        cp = np.array([(a, b) for si, sj in need for a in cart_of_shell[si] for b in cart_of_shell[sj]], dtype=np.int64)
    pi = np.ascontiguousarray(cp[:, 0], dtype=np.int32); pj = np.ascontiguousarray(cp[:, 1], dtype=np.int32)
    out = np.zeros((len(cp), auxmol.ncart))
    lib.gto_eri_3c_pairs(mol.ncart, mol._arr(), auxmol.ncart, auxmol._arr(), C.c_long(len(cp)), pi.ctypes.data_as(C.c_void_p),
                         pj.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if not auxmol.cart:
        out = out @ auxmol.c2s
    if mol.cart:
        return out
    where = {(int(a), int(b)): k for k, (a, b) in enumerate(cp)}
    res = np.zeros((len(pairs), out.shape[1]))
    for k, (p, q) in enumerate(pairs):
        ca, cb = np.nonzero(mol.c2s[:, p])[0], np.nonzero(mol.c2s[:, q])[0]
        for a in ca:
            for b in cb:
                res[k] += mol.c2s[a, p] * mol.c2s[b, q] * out[where[(int(a), int(b))]]
    return res


class RHF:
    """Closed-shell RHF with DIIS in the AO basis (generalised eigenproblem through S^-1/2)."""

    def __init__(self, mol: Mole, conv_tol=1e-12, max_cycle=100):
        self.mol = mol
        self.conv_tol, self.max_cycle = conv_tol, max_cycle
        self.mo_coeff = self.mo_energy = self.mo_occ = None
        self.e_tot = None
        self.converged = False
        self._eri = None
        self._S = self._h = None

    def get_ovlp(self):
        if self._S is None:
            S, T, V = self.mol.one_electron()
            self._S, self._h = S, T + V
        return self._S

    def get_hcore(self):
        self.get_ovlp()
        return self._h.copy()

    def energy_nuc(self):
        return self.mol.energy_nuc()

    def _jk(self, dm):
        e = self._eri
        return np.einsum("pqrs,rs->pq", e, dm, optimize=True), np.einsum("pqrs,qs->pr", e, dm, optimize=True)

    def get_veff(self, dm=None):
        dm = self.make_rdm1() if dm is None else dm
        J, K = self._jk(dm)
        return J - 0.5 * K

    def make_rdm1(self):
        no = self.mol.nelectron // 2
        return 2.0 * self.mo_coeff[:, :no] @ self.mo_coeff[:, :no].T

    def kernel(self):
        S = self.get_ovlp(); h = self._h
        if self._eri is None:
            self._eri = self.mol.eri_s1()
        no = self.mol.nelectron // 2
        w, U = np.linalg.eigh(S)
        X = U / np.sqrt(w) @ U.T
        e, c = np.linalg.eigh(X @ h @ X)
        Cm = X @ c
        dm = 2.0 * Cm[:, :no] @ Cm[:, :no].T
        fs, es = [], []
        e_old = None
        for cyc in range(self.max_cycle):
            J, K = self._jk(dm)
            F = h + J - 0.5 * K
            e_el = 0.5 * np.sum((h + F) * dm)
            err = X @ (F @ dm @ S - S @ dm @ F) @ X
            if e_old is not None and abs(e_el - e_old) < self.conv_tol and np.linalg.norm(err) < 1e-8:
                self.converged = True
                break
            e_old = e_el
            fs.append(F); es.append(err); fs, es = fs[-8:], es[-8:]
            Fd = F
            if len(fs) > 1:
                m = len(fs)
                B = np.zeros((m + 1, m + 1)); B[-1, :] = B[:, -1] = 1.0; B[-1, -1] = 0.0
                for i in range(m):
                    for j in range(m):
                        B[i, j] = np.vdot(es[i], es[j])
                rhs = np.zeros(m + 1); rhs[-1] = 1.0
                try:
                    cf = np.linalg.solve(B, rhs)[:m]
                    Fd = sum(a * b for a, b in zip(cf, fs))
                except np.linalg.LinAlgError:
                    pass
            e, c = np.linalg.eigh(X @ Fd @ X)
            Cm = X @ c
            dm = 2.0 * Cm[:, :no] @ Cm[:, :no].T
        J, K = self._jk(dm)
        F = h + J - 0.5 * K
        e, c = np.linalg.eigh(X @ F @ X)
        self.mo_energy, self.mo_coeff = e, X @ c
        self.mo_occ = np.zeros(len(e)); self.mo_occ[:no] = 2.0
        self.e_tot = 0.5 * np.sum((h + F) * dm) + self.energy_nuc()
        return self.e_tot
