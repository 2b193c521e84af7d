"""Synthetic molecular DF integral source (data only; shared by tests/golden/make_golden_df.py, the oracle test and the device test).

What PySCF / libcint would supply to molbe/eri_onthefly.py:45-145, generated from a seed with the symmetries of the real quantities:
(mu nu|P) symmetric in mu, nu, grouped in auxiliary SHELLS of unequal size (the reference blocks its loop over shells, :114-119), and a
symmetric positive definite metric (P|Q)."""
import numpy as np


class SyntheticDFSource:
    def __init__(self, nao, aux_shell_sizes, seed):
        rng = np.random.default_rng(seed)
        self.nao = nao
        self.aux_shell_sizes = list(aux_shell_sizes)
        self.naux = sum(self.aux_shell_sizes)
        self.aux_ao_loc = np.concatenate([[0], np.cumsum(self.aux_shell_sizes)]).astype(int)
        a = 0.3 * rng.standard_normal((nao, nao, self.naux))
        self.pqL = np.ascontiguousarray(0.5 * (a + a.transpose(1, 0, 2)))           # (mu nu|P), the layout getints3c returns
        A = rng.standard_normal((self.naux, self.naux))
        self.j2c = A @ A.T / self.naux + 0.5 * np.eye(self.naux)

    def block(self, s0, s1):
        """(mu nu|P) for the auxiliary shells [s0, s1)"""
        return np.array(self.pqL[:, :, self.aux_ao_loc[s0]:self.aux_ao_loc[s1]])


def fragment_TAs(nao, ns, seed):
    rng = np.random.default_rng(seed)
    out = []
    for n in ns:
        Q, _ = np.linalg.qr(rng.standard_normal((nao, nao)))
        out.append(np.ascontiguousarray(Q[:, :n]))
    return out


CASES = {
    # name: (nao, auxiliary shell sizes, seed, fragment sizes, shells per block or None for one block)
    "small": (7, (1, 3, 5, 1, 3), 21, (5, 4), 2),
    "medium": (16, (1, 1, 3, 3, 5, 5, 7, 1, 3, 5), 22, (10, 7, 12), 3),
    "oneblock": (9, (3, 5, 7), 23, (6,), None),
}


def make_case(name):
    nao, shells, seed, ns, step = CASES[name]
    return SyntheticDFSource(nao, shells, seed), fragment_TAs(nao, ns, seed + 100), step
