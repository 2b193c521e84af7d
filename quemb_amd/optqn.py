"""Broyden quasi-Newton solver of the density-matching equations (host; 'next' row f.1 of SURVEY section 8).

Mirror of shared/external/optqn.py: `FrankQN` (:158-247) with the Li-Fukushima derivative-free line search
(`line_search_LF`, :25-59) or the Broyden trust-region dog-leg step (`trustRegion`, :62-155).  The state is the
inverse-Jacobian approximation updated with the good-Broyden (Sherman-Morrison) formula; the reference keeps the
same update twice (explicit `Binv` for the trust region and the recursive `get_Bnfn` for the line search),
which are algebraically identical, so one explicit matrix is kept here.  N_pot is O(10^2-10^3): host NumPy.
"""

from __future__ import annotations

import numpy as np
from numpy.linalg import inv, norm, pinv


def line_search_LF(func, xold, fold, dx, iter_, verbose=True):
    """D.-H. Li and M. Fukushima, Optim. Methods Softw. 13, 181 (2000): accept x + dx when the residual drops
    enough, else back-track alpha *= 0.1 (at most 20 evaluations)."""
    beta, rho, sigma1, sigma2 = 0.1, 0.9, 1e-3, 1e-3
    eta = (iter_ + 1) ** -2.0
    alp = 1.0
    xk = xold + dx
    fk = func(xk)
    nev = 1
    ndx, nfk, nfo = norm(dx), norm(fk), norm(fold)
    if nfk > rho * nfo - sigma2 * ndx ** 2.0:
        while nfk > (1.0 + eta) * nfo - sigma1 * alp ** 2.0 * ndx ** 2.0:
            alp *= beta
            xk = xold + alp * dx
            fk = func(xk)
            nev += 1
            nfk = norm(fk)
            if nev == 20:
                break
    if verbose:
        print(" No. of line search steps in QN opt :", nev, flush=True)
    return alp, xk, fk


def trustRegion(func, xold, fold, Binv, c=0.5, verbose=True):
    """Broyden trust-region dog-leg step (reference optqn.py:62-155; IAENG IJCS 46(3) 2019, Algorithm 1)."""
    p = 0
    rho = 0.001
    ratio = 0.0
    B = inv(Binv)
    dx_gn = -(Binv @ Binv.T) @ B.T @ fold
    dx_sd = -B.T @ fold
    t = norm(dx_sd) ** 2 / norm(B @ dx_sd) ** 2
    prev = None
    ared = 0.0
    fnew = fold
    while ratio < rho or ared < 0.0:
        radius = c ** p
        scale = max(1.0, norm(xold)) * radius
        if norm(dx_gn) < scale:
            kind, dx = "Gauss-Newton", dx_gn
        elif t * norm(dx_sd) > scale:
            kind, dx = "Steepest Descent", radius / norm(dx_sd) * dx_sd
        else:
            kind = "Dog Leg"
            tdx = t * dx_sd
            diff = dx_gn - tdx
            s = 1
            dx = tdx + s * diff
            while norm(dx) > radius and s > 0:
                s -= 0.001
                dx = tdx + s * diff
        if verbose:
            print("  Trust Region Optimization Step ", p, ":", kind, flush=True)
        if prev is None or not np.all(dx == prev):
            fnew = func(xold + dx)
            ared = 0.5 * (norm(fold) ** 2 - norm(fnew) ** 2)
            pred = 0.5 * (norm(fold) ** 2 - norm(fold + B @ dx) ** 2)
        ratio = ared / pred
        p += 1
        prev = dx
    return xold + dx, fnew


class FrankQN:
    """x_{k+1} = x_k - alpha B_k f_k with B_k the Broyden inverse Jacobian, B_0 = pinv(J0)."""

    def __init__(self, func, x0, f0, J0, trust=0.5, max_space=500, verbose=True):
        self.func = func
        self.x0 = np.asarray(x0, dtype=float)
        self.f0 = f0
        self.n = self.x0.size
        self.B0 = pinv(J0)
        self.Binv = None
        self.trust = trust
        self.max_subspace = max_space
        self.xnew = self.xold = self.fnew = self.fold = None
        self.verbose = verbose

    def next_step(self, iter, trust_region=False):
        if iter == 0:
            self.xnew = self.x0
            self.fnew = self.func(self.xnew) if self.f0 is None else self.f0
            self.Binv = self.B0.copy()
        else:
            dx = self.xnew - self.xold
            df = self.fnew - self.fold
            self.Binv += np.outer(dx - self.Binv @ df, dx @ self.Binv) / (dx @ self.Binv @ df)
        self.xold = self.xnew.copy()
        self.fold = self.fnew.copy()
        if trust_region:
            self.xnew, self.fnew = trustRegion(self.func, self.xold, self.fold, self.Binv, c=self.trust, verbose=self.verbose)
        else:
            step = self.Binv @ self.fold
            _, self.xnew, self.fnew = line_search_LF(self.func, self.xold, self.fold, -step, iter, verbose=self.verbose)
