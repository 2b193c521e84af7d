"""Oracle for the "next" row f.3: CCSD Lambda equations and the relaxed (response) density matrices that
solve_ccsd(..., relax=True) takes from PySCF (molbe/solver.py:925-939: `mycc.make_rdm1()`, `make_rdm2(mycc, t1, t2,
mycc.l1, mycc.l2, with_dm1=not use_cumulant)`).  Test infrastructure.

PySCF (cc/ccsd_lambda.py, cc/ccsd_rdm.py) is not installed in this image and the reference holds no golden value for
relax_density=True, so this restatement does not copy PySCF's intermediates; it uses their DEFINITION.  With the CC
Lagrangian

    L(t, z; f, V) = E(t; f, V) + sum_mu z_mu r_mu(t; f, V),       r = amplitude_numerators(t) - D o t   (ccsd.py)

the Lambda equations are dL/dt = 0 and the response densities are dL/df (1-particle, correlation part) and dL/dV
(2-particle, normal ordered with respect to the HF determinant == PySCF's `with_dm1=False`).  The derivatives are taken
by reverse-mode differentiation of the oracle's own amplitude equations (a small tape over numpy.einsum below).
Multipliers z differ from PySCF's l1, l2 by a normalisation that cancels in the densities.

Pinned by tests/test_oracle_lambda.py: (i) dE_corr/df and dE_corr/dV by finite differences of the re-converged CCSD
energy, (ii) two-electron systems, where CCSD is exact: 1-RDM == FCI 1-RDM, (iii) z = 0 reproduces the unrelaxed
rdm.py expressions.  Parity against PySCF itself: unpinned.
"""
import numpy as np

from . import ccsd as occsd


class Var:
    """A node of the reverse-mode tape: value + [(parent, vjp)]."""
    __array_priority__ = 1000.0

    def __init__(self, v, parents=()):
        self.v = np.asarray(v, dtype=float)
        self.parents = tuple(parents)

    @property
    def shape(self):
        return self.v.shape

    def _lin(self, other, sa, sb):
        if isinstance(other, Var):
            return Var(sa * self.v + sb * other.v, [(self, lambda g: sa * g), (other, lambda g: sb * g)])
        return Var(sa * self.v + sb * np.asarray(other), [(self, lambda g: sa * g)])

    def __add__(self, o): return self._lin(o, 1.0, 1.0)
    __radd__ = __add__
    def __sub__(self, o): return self._lin(o, 1.0, -1.0)
    def __rsub__(self, o): return self._lin(o, -1.0, 1.0)
    def __neg__(self): return Var(-self.v, [(self, lambda g: -g)])

    def __mul__(self, c):
        if isinstance(c, Var):
            raise TypeError("products of tape variables go through es()")
        c = float(c)
        return Var(c * self.v, [(self, lambda g: c * g)])
    __rmul__ = __mul__

    def transpose(self, *axes):
        axes = axes[0] if len(axes) == 1 and not isinstance(axes[0], int) else axes
        inv = np.argsort(axes)
        return Var(self.v.transpose(axes), [(self, lambda g: g.transpose(inv))])

    def copy(self):
        return self

    def __getitem__(self, idx):
        def vjp(g, idx=idx, shape=self.v.shape):
            z = np.zeros(shape); z[idx] = g
            return z
        return Var(self.v[idx], [(self, vjp)])


def es(subs, *ops):
    """numpy.einsum that records the vector-Jacobian products of its tape operands."""
    vals = [o.v if isinstance(o, Var) else np.asarray(o) for o in ops]
    out = np.einsum(subs, *vals, optimize=True)
    ins, outsub = subs.split("->")
    ins = ins.split(",")
    parents = []
    for k, o in enumerate(ops):
        if not isinstance(o, Var):
            continue
        rest = [m for m in range(len(ops)) if m != k]
        sub = ",".join([outsub] + [ins[m] for m in rest]) + "->" + ins[k]
        parents.append((o, lambda g, sub=sub, rest=rest: np.einsum(sub, g, *[vals[m] for m in rest], optimize=True)))
    return Var(out, parents) if parents else out


def backward(outputs, seeds):
    """Accumulate cotangents from `outputs` (seeded with `seeds`) to every node; returns {id(node): cotangent}."""
    order, seen = [], set()

    def visit(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for p, _ in n.parents:
            visit(p)
        order.append(n)
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    for o in outputs:
        visit(o)
    grad = {}
    for o, s in zip(outputs, seeds):
        grad[id(o)] = grad.get(id(o), 0.0) + np.asarray(s, dtype=float)
    for n in reversed(order):
        g = grad.get(id(n))
        if g is None:
            continue
        for p, vjp in n.parents:
            c = vjp(g)
            grad[id(p)] = c if id(p) not in grad else grad[id(p)] + c
    return grad


class _TapeEris:
    BLOCKS = ("oooo", "ovoo", "ovov", "oovv", "ovvo", "ovvv", "vvvv")

    def __init__(self, eris):
        self.nocc, self.nmo, self.mo_energy = eris.nocc, eris.nmo, eris.mo_energy
        self.fock = Var(eris.fock)
        for b in self.BLOCKS:
            setattr(self, b, Var(getattr(eris, b)))


class Lagrangian:
    """Tape of E and r at fixed (t1, t2); `vjp(z1, z2)` differentiates L = E + z.r with respect to t, f and the ERI blocks."""

    def __init__(self, t1, t2, eris):
        self.nocc, self.nvir = t1.shape
        self.te = _TapeEris(eris)
        self.t1, self.t2 = Var(t1), Var(t2)
        eo, ev = eris.mo_energy[: self.nocc], eris.mo_energy[self.nocc:]
        self.eia = eo[:, None] - ev[None, :]
        self.eijab = self.eia[:, None, :, None] + self.eia[None, :, None, :]
        n1, n2 = occsd.amplitude_numerators(self.t1, self.t2, self.te, es=es)
        self.r1 = n1            # r = n - D o t: the -D o t part is added explicitly in vjp()
        self.r2 = n2
        self.E = occsd.energy(self.t1, self.t2, self.te, es=es)

    def residual_norm(self):
        return float(np.sqrt(np.linalg.norm(self.r1.v - self.eia * self.t1.v) ** 2 + np.linalg.norm(self.r2.v - self.eijab * self.t2.v) ** 2))

    def vjp(self, z1, z2):
        """Cotangents of L = E + z1.(n1 - D t1) + z2.(n2 - D t2): returns dict with t1, t2, fock and every ERI block."""
        g = backward([self.E, self.r1, self.r2], [1.0, z1, z2])
        out = {"t1": g[id(self.t1)] - self.eia * z1, "t2": g[id(self.t2)] - self.eijab * z2, "fock": g[id(self.te.fock)]}
        for b in _TapeEris.BLOCKS:
            out[b] = g.get(id(getattr(self.te, b)), np.zeros(getattr(self.te, b).shape))
        return out


def solve_lambda(t1, t2, eris, conv_tol=1e-10, max_cycle=200, diis_space=6):
    """Fixed-point iteration z <- (dE/dt + (dn/dt)^T z) / D  (the transpose of the amplitude iteration), DIIS as in ccsd.kernel.
    Returns (z1, z2, n_iter, lagrangian)."""
    lag = Lagrangian(t1, t2, eris)
    o, v = t1.shape
    z1 = np.zeros_like(t1); z2 = np.zeros_like(t2)
    adiis = occsd.DIIS(diis_space)
    for it in range(1, max_cycle + 1):
        g = lag.vjp(z1, z2)
        # dL/dt = g_t (includes -D z); stationarity: z = (g_t + D z) / D
        z1n = (g["t1"] + lag.eia * z1) / lag.eia
        z2n = (g["t2"] + lag.eijab * z2) / lag.eijab
        dz = np.sqrt(np.linalg.norm(z1n - z1) ** 2 + np.linalg.norm(z2n - z2) ** 2)
        vec = adiis.update(np.concatenate([z1n.ravel(), z2n.ravel()]))
        z1 = vec[: o * v].reshape(o, v); z2 = vec[o * v:].reshape(o, o, v, v)
        if dz < conv_tol:
            break
    else:
        raise RuntimeError("oracle Lambda equations did not converge")
    return z1, z2, it, lag


def response_densities(lag, z1, z2):
    """(dm1, V_bar): dm1 = MO-basis 1-RDM including the HF part (== mycc.make_rdm1()), V_bar[p,q,r,s] = dL/d(pq|rs) scattered
    from the block cotangents (not symmetrised; E2 = sum V_bar o (pq|rs)).  z = 0 gives the unrelaxed quantities."""
    g = lag.vjp(z1, z2)
    o, n = lag.nocc, lag.nocc + lag.nvir
    fbar = g["fock"]
    dm1 = 0.5 * (fbar + fbar.T)
    dm1[np.diag_indices(o)] += 2.0
    V = np.zeros((n,) * 4)
    O, Vv = slice(0, o), slice(o, n)
    V[O, O, O, O] += g["oooo"]; V[O, Vv, O, O] += g["ovoo"]; V[O, Vv, O, Vv] += g["ovov"]; V[O, O, Vv, Vv] += g["oovv"]
    V[O, Vv, Vv, O] += g["ovvo"]; V[O, Vv, Vv, Vv] += g["ovvv"]; V[Vv, Vv, Vv, Vv] += g["vvvv"]
    return dm1, V


def symmetrise8(V):
    V = 0.5 * (V + V.transpose(1, 0, 2, 3))
    V = 0.5 * (V + V.transpose(0, 1, 3, 2))
    return 0.5 * (V + V.transpose(2, 3, 0, 1))


def make_rdm2_relaxed(lag, z1, z2):
    """8-fold symmetrised normal-ordered 2-RDM Gamma with E2 = 1/2 sum Gamma o (pq|rs) -- the part of PySCF's
    make_rdm2(..., with_dm1=False) that any contraction with the (8-fold symmetric) ERIs can see, which is all that
    get_frag_energy (molbe/helper.py:307-321) uses."""
    _, V = response_densities(lag, z1, z2)
    return 2.0 * symmetrise8(V)
