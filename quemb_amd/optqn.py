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


# Li-Fukushima acceptance test, constants of the reference's choice (optqn.py:25-59): shrink factor of the step, sufficient-decrease
# fraction of the residual, and the two weights of the squared step length
_LF_SHRINK, _LF_DECREASE, _LF_W_LOOP, _LF_W_FIRST, _LF_MAX_EVALS = 0.1, 0.9, 1e-3, 1e-3, 20


def line_search_LF(func, xold, fold, dx, iter_, verbose=True):
    """Derivative-free back-tracking of D.-H. Li and M. Fukushima, Optim. Methods Softw. 13, 181 (2000).  The full quasi-Newton step is
    taken when |f(x + dx)| <= 0.9 |f(x)| - 1e-3 |dx|^2; otherwise the step length is cut by ten until
    |f(x + a dx)| <= (1 + (k + 1)^-2) |f(x)| - 1e-3 a^2 |dx|^2 holds or twenty residuals have been evaluated.
    Returns (step length, x, f(x)) -- same sequence of trial points as the reference's routine of this name."""
    slack = (iter_ + 1) ** -2.0                       # forcing term of outer iteration k: the test relaxes as 1 / (k + 1)^2
    step_len2 = norm(dx) ** 2.0
    res_old = norm(fold)
    length = 1.0
    trial = xold + dx
    f_trial = func(trial)
    evals = 1
    res_new = norm(f_trial)
    full_step_ok = not (res_new > _LF_DECREASE * res_old - _LF_W_FIRST * step_len2)
    while not full_step_ok and evals < _LF_MAX_EVALS:
        if not (res_new > (1.0 + slack) * res_old - _LF_W_LOOP * length ** 2.0 * step_len2):
            break
        length *= _LF_SHRINK
        trial = xold + length * dx
        f_trial = func(trial)
        evals += 1
        res_new = norm(f_trial)
    if verbose:
        print(f" quasi-Newton line search: {evals} residual evaluation(s), step length {length:g}", flush=True)
    return length, trial, f_trial


def trustRegion(func, xold, fold, Binv, c=0.5, verbose=True):
    """Dog-leg step inside a shrinking trust region on the Broyden model (the reference's routine of this name, optqn.py:62-155, after
    IAENG IJCS 46(3) 2019, Algorithm 1).  Radius c^p times max(1, |x|), p = 0, 1, ...: the Gauss-Newton point if it lies inside, the scaled
    steepest-descent direction if even the Cauchy point lies outside, else the point of the dog leg between them that reaches the radius
    (found by walking back from the Gauss-Newton end in steps of 1/1000).  A step is accepted when the actual reduction of |f|^2 / 2 is
    positive and at least 0.001 of the reduction the linear model predicts."""
    accept_ratio = 0.001
    jac = inv(Binv)
    newton_pt = -(Binv @ Binv.T) @ jac.T @ fold
    descent = -jac.T @ fold
    cauchy_len = norm(descent) ** 2 / norm(jac @ descent) ** 2
    half_res_old = 0.5 * norm(fold) ** 2
    last_step = None
    gain_ratio, actual = 0.0, 0.0
    f_new = fold
    shrinks = 0
    while gain_ratio < accept_ratio or actual < 0.0:
        radius = c ** shrinks
        reach = max(1.0, norm(xold)) * radius
        if norm(newton_pt) < reach:
            label, step = "Gauss-Newton point", newton_pt
        elif cauchy_len * norm(descent) > reach:
            label, step = "steepest descent to the radius", radius / norm(descent) * descent
        else:
            label = "dog leg"
            cauchy_pt = cauchy_len * descent
            leg = newton_pt - cauchy_pt
            frac = 1
            step = cauchy_pt + frac * leg
            while norm(step) > radius and frac > 0:
                frac -= 0.001
                step = cauchy_pt + frac * leg
        if verbose:
            print(f"  trust region, radius {radius:g}: {label}", flush=True)
        if last_step is None or not np.all(step == last_step):
            f_new = func(xold + step)
            actual = half_res_old - 0.5 * norm(f_new) ** 2
            predicted = half_res_old - 0.5 * norm(fold + jac @ step) ** 2
        gain_ratio = actual / predicted
        shrinks += 1
        last_step = step
    return xold + step, f_new


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
