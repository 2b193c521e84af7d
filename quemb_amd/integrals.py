"""Integral source + mean-field for real molecules without PySCF.

`Mole` / `RHF` expose the handful of PySCF attributes `BE` reads (molbe/mbe.py:361-373): `mol.nelectron`,
`mo_coeff`, `mo_energy`, `e_tot`, `_eri`, `energy_nuc()`, `get_hcore()`, `get_ovlp()`, `make_rdm1()`,
`get_veff()`.  Integrals come from the in-tree host library libqemb_gto.so (csrc_host/gto_ints.c, s and p
Cartesian Gaussians, McMurchie-Davidson); basis: STO-3G for H and C (the reference's test systems).
This is upstream of the hot path -- CPU work in the reference too (libcint) -- and exists so that the reference's
end-to-end golden energies can be reproduced from first principles.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

BOHR = 0.52917721092            # Angstrom per Bohr (pyscf.data.nist.BOHR)
_HERE = Path(__file__).resolve().parent
GTO_LIB = _HERE / "libqemb_gto.so"
MAXPRIM = 8

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


class Mole:
    def __init__(self, atom, basis="sto-3g", unit="Angstrom"):
        if basis.lower() not in _BASES:
            raise NotImplementedError("built-in bases: STO-3G (H, C), cc-pVDZ (H)")
        table = _BASES[basis.lower()]
        if isinstance(atom, (str, Path)):
            atom = read_xyz(atom)
        scale = 1.0 / BOHR if unit.lower().startswith("a") else 1.0
        self.atom = [(sym, tuple(scale * np.asarray(xyz, dtype=float))) for sym, xyz in atom]
        self.nelectron = sum(_Z[s] for s, _ in self.atom)
        self.bfs = []
        self.ao_atom = []
        for ia, (sym, xyz) in enumerate(self.atom):
            if sym not in table:
                raise NotImplementedError(f"no {basis} basis for {sym} is built in")
            for l, exps, coefs in table[sym]:
                comps = [(0, 0, 0)] if l == "s" else [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
                for lmn in comps:
                    self.bfs.append(self._make_bf(xyz, lmn, exps, coefs))
                    self.ao_atom.append(ia)
        self.nao = len(self.bfs)

    @staticmethod
    def _make_bf(xyz, lmn, exps, coefs):
        L = sum(lmn)
        e = np.asarray(exps, dtype=float); c = np.asarray(coefs, dtype=float)
        norm = (2 * e / np.pi) ** 0.75 * (4 * e) ** (L / 2.0) / np.sqrt(_dfact(2 * lmn[0] - 1) * _dfact(2 * lmn[1] - 1) * _dfact(2 * lmn[2] - 1))
        cc = c * norm
        # renormalise the contraction to unit self-overlap
        pref = np.pi ** 1.5 * _dfact(2 * lmn[0] - 1) * _dfact(2 * lmn[1] - 1) * _dfact(2 * lmn[2] - 1) / 2.0 ** L
        s = sum(cc[i] * cc[j] * pref / (e[i] + e[j]) ** (L + 1.5) for i in range(len(e)) for j in range(len(e)))
        cc = cc / np.sqrt(s)
        b = _BF()
        b.ctr[:] = xyz; b.lmn[:] = lmn; b.nprim = len(e)
        for i in range(len(e)):
            b.ex[i] = e[i]; b.co[i] = cc[i]
        return b

    def _arr(self):
        return (_BF * self.nao)(*self.bfs)

    @property
    def natm(self):
        return len(self.atom)

    def atom_charge(self, ia):
        return _Z[self.atom[ia][0]]

    def aoslice_by_atom(self):
        """PySCF's (shell0, shell1, ao0, ao1) per atom; only the AO range is meaningful here (shells are not tracked)."""
        ao = np.asarray(self.ao_atom)
        out = []
        for ia in range(self.natm):
            w = np.nonzero(ao == ia)[0]
            out.append((0, 0, int(w[0]), int(w[-1]) + 1))
        return np.array(out)

    def energy_nuc(self):
        e = 0.0
        for i, (si, xi) in enumerate(self.atom):
            for sj, xj in self.atom[:i]:
                e += _Z[si] * _Z[sj] / np.linalg.norm(np.asarray(xi) - np.asarray(xj))
        return e

    def one_electron(self):
        lib = _load()
        n = self.nao
        S = np.zeros((n, n)); T = np.zeros((n, n)); V = np.zeros((n, n))
        xyz = np.ascontiguousarray([a[1] for a in self.atom], dtype=float)
        Z = np.ascontiguousarray([_Z[a[0]] for a in self.atom], dtype=float)
        lib.gto_one_electron(n, self._arr(), len(self.atom), xyz.ctypes.data_as(C.c_void_p), Z.ctypes.data_as(C.c_void_p),
                             S.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
        return S, T, V

    def eri_s1(self):
        lib = _load()
        n = self.nao
        out = np.zeros((n, n, n, n))
        lib.gto_eri_s1(n, self._arr(), out.ctypes.data_as(C.c_void_p))
        return out

    def intor(self, name):
        if name == "int1e_ovlp":
            return self.one_electron()[0]
        raise NotImplementedError(name)


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
