"""BEOPT -- outer density-matching / chemical-potential optimisation (mirror of molbe/opt.py:22-213).

`objfunc(pot)` is one fragment sweep (be_func or its multi-GPU twin); the quasi-Newton update itself is host
NumPy (`optqn.FrankQN`).  On several GPUs every rank runs the same optimiser on the same all-reduced residual,
so the potentials stay bit-identical across ranks without a broadcast.
"""

from __future__ import annotations

import warnings

import numpy as np

from .optqn import FrankQN


class BEOPT:
    def __init__(self, pot, Fobjs, Nocc, enuc, scratch_dir=None, solver="CCSD", nproc=1, ompnum=1, only_chem=False,
                 use_cumulant=True, max_space=500, conv_tol=1.0e-6, relax_density=False, ebe_hf=0.0, solver_args=None, *,
                 sweep=None, verbose=True):
        self.pot = list(pot)
        self.Fobjs, self.Nocc, self.enuc = Fobjs, Nocc, enuc
        self.solver, self.only_chem, self.use_cumulant = solver, only_chem, use_cumulant
        self.max_space, self.conv_tol, self.relax_density, self.ebe_hf = max_space, conv_tol, relax_density, ebe_hf
        self.iter = 0
        self.err = 0.0
        self.Ebe = np.array([[0.0]])
        self.verbose = verbose
        if sweep is None:
            from .solver import be_func

            def sweep(p, **kw):
                return be_func(p, Fobjs, Nocc, solver, enuc, **kw)
        self._sweep = sweep
        self.n_objfunc = 0

    def objfunc(self, xk):
        """molbe/opt.py:89-144: error vector of one sweep; records the RMS error and the BE energies."""
        err_, errvec_, ebe_ = self._sweep(list(xk), only_chem=self.only_chem, use_cumulant=self.use_cumulant, eeval=True,
                                          return_vec=True, relax_density=self.relax_density)
        self.err = err_
        self.Ebe = ebe_
        self.pot = list(xk)
        self.n_objfunc += 1
        return errvec_

    def _say(self, *a):
        if self.verbose:
            print(*a, flush=True)

    def optimize(self, method, J0=None, trust_region=False):
        """molbe/opt.py:146-213."""
        if method != "QN":
            raise ValueError("This optimization method for BE is not supported")
        self._say("-- Beginning optimization iteration ", self.iter)
        f0 = self.objfunc(self.pot)
        self._say(f"Error in density matching      :   {self.err:>2.4e}")
        optQN = FrankQN(self.objfunc, np.array(self.pot), f0, J0, max_space=self.max_space, verbose=self.verbose)
        if self.err < self.conv_tol:
            self._say("CONVERGED w/o Optimization Steps")
            return
        for _ in range(self.max_space):
            self._say("-- In iter ", self.iter)
            optQN.next_step(self.iter, trust_region=trust_region)
            self.iter += 1
            self._say(f"Error in density matching      :   {self.err:>2.4e}")
            if self.err < self.conv_tol:
                self._say("CONVERGED")
                break
        if self.err >= self.conv_tol:
            warnings.warn(f"BE DID NOT CONVERGE IN {self.max_space} STEPS")
