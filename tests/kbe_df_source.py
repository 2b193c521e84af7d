"""Synthetic Gamma-point CC-GDF integral source (data only; shared by the golden generator, the oracle tests and the device tests).

What PySCF-PBC would supply to kbe/eri_onthefly.py:48-241, generated from a seed with the symmetries the real quantities have:
(G|mu nu) symmetric in mu, nu and conjugate under G -> -G; ft_aux(-G) = conj(ft_aux(G)); real symmetric real-space blocks;
a symmetric metric, positive definite or (case "indefinite") with one negative and one numerically zero eigenvalue.  `imag_break`
adds a conjugation-symmetry-breaking term to the plane-wave blocks (the reference's imaginary-part test then has something to see)."""
import numpy as np


class SyntheticGammaSource:
    def __init__(self, nao, naux, nhalf, seed, indefinite=False, imag_break=0.0, aux_shell_sizes=None):
        rng = np.random.default_rng(seed)
        self.nao, self.naux = nao, naux
        self.n_planewaves = 2 * nhalf + 1
        sym = lambda a: 0.5 * (a + a.transpose(0, 2, 1))
        half = 0.2 * sym(rng.standard_normal((nhalf, nao, nao)) + 1j * rng.standard_normal((nhalf, nao, nao)))
        g0 = 0.3 * sym(rng.standard_normal((1, nao, nao))).astype(np.complex128)
        # order: G = 0, then (+G_k, -G_k) pairs
        blocks = [g0[0]]
        for k in range(nhalf):
            blocks += [half[k], half[k].conj()]
        self._pw = np.array(blocks)
        if imag_break:
            self._pw = self._pw + imag_break * sym(rng.standard_normal(self._pw.shape)) * 1j
        fh = 0.3 * (rng.standard_normal((nhalf, naux)) + 1j * rng.standard_normal((nhalf, naux)))
        f0 = 0.3 * rng.standard_normal((1, naux)).astype(np.complex128)
        rows = [f0[0]]
        for k in range(nhalf):
            rows += [fh[k], fh[k].conj()]
        self._ft = np.array(rows)
        self._rs = 0.25 * sym(rng.standard_normal((naux, nao, nao)))
        A = rng.standard_normal((naux, naux))
        Q, _ = np.linalg.qr(A)
        d = np.linspace(0.4, 2.5, naux)
        if indefinite:
            d[0], d[1] = -3e-3, 1e-17
        self._j2c = (Q * d) @ Q.T
        self._j2c = 0.5 * (self._j2c + self._j2c.T)
        # auxiliary shells (the reference blocks the real-space loop over SHELLS, :201-206)
        self.aux_shell_sizes = list(aux_shell_sizes) if aux_shell_sizes is not None else [1] * naux
        assert sum(self.aux_shell_sizes) == naux
        self.aux_ao_loc = np.concatenate([[0], np.cumsum(self.aux_shell_sizes)])

    def j2c(self):
        return self._j2c

    def pw_block(self, g0, g1):
        return self._pw[g0:g1]

    def ft_aux_block(self, g0, g1):
        return self._ft[g0:g1]

    def rs_block(self, p0, p1):
        return self._rs[p0:p1]


class GammaSourceFromFactor:
    """a Gamma-point CC-GDF source whose fitted tensor is a given DF factor B (naux, N, N): unit metric, real-space blocks = B, and a
    plane-wave part that cancels between +G and -G up to a real remainder folded into the blocks (exercises add_pw_block)"""

    def __init__(self, B, seed=3):
        rng = np.random.default_rng(seed)
        self.naux, self.nao = B.shape[0], B.shape[1]
        sym = lambda a: 0.5 * (a + a.transpose(0, 2, 1))
        half = 0.1 * sym(rng.standard_normal((2, self.nao, self.nao)) + 1j * rng.standard_normal((2, self.nao, self.nao)))
        self._pw = np.array([half[0], half[0].conj(), half[1], half[1].conj()])
        fh = 0.2 * (rng.standard_normal((2, self.naux)) + 1j * rng.standard_normal((2, self.naux)))
        self._ft = np.array([fh[0], fh[0].conj(), fh[1], fh[1].conj()])
        pw_part = np.einsum("GL,Gpq->Lpq", self._ft.conj(), self._pw)
        assert np.abs(pw_part.imag).max() < 1e-14
        self._rs = B - pw_part.real
        self.n_planewaves = 4

    def j2c(self): return np.eye(self.naux)
    def pw_block(self, g0, g1): return self._pw[g0:g1]
    def ft_aux_block(self, g0, g1): return self._ft[g0:g1]
    def rs_block(self, p0, p1): return self._rs[p0:p1]


def fragment_TAs(nao, ns, seed):
    rng = np.random.default_rng(seed)
    out = []
    for n in ns:
        Q, _ = np.linalg.qr(rng.standard_normal((nao, nao)))
        out.append(np.ascontiguousarray(Q[:, :n]))
    return out


CASES = {
    # name: (nao, naux, nhalf, seed, kwargs, fragment sizes, expected to raise)
    "pd": (6, 10, 4, 11, dict(aux_shell_sizes=[1, 3, 1, 5]), (4, 3), False),
    "indefinite": (7, 12, 5, 12, dict(indefinite=True, aux_shell_sizes=[3, 3, 1, 5]), (5, 4), False),
    "small_imag": (6, 9, 3, 13, dict(imag_break=1e-8), (4,), False),
    "large_imag": (5, 8, 3, 14, dict(imag_break=5e-2), (3,), True),
}


def make_case(name):
    nao, naux, nhalf, seed, kw, ns, raises = CASES[name]
    return SyntheticGammaSource(nao, naux, nhalf, seed, **kw), fragment_TAs(nao, ns, seed + 100), raises
