"""Numerical Jacobian of the BE matching conditions (molbe/numerical_jac.py:11-187), `BE.optimize(jac_solver="Numerical")`.

Column k of J0 is the central difference of the error vector with respect to potential k.  The chemical-potential column takes
two full sweeps; every other potential lives in ONE fragment, so its column takes two solves of that fragment only (the
reference does the same, numerical_jac.py:100-168).  The error vector is linear in the fragments' 1-RDMs, hence the difference
of the two error vectors is the matching map applied to rdm1(+h) - rdm1(-h) of that one fragment: with fragments sharded over
ranks each rank fills the columns of its own fragments and one all-reduce assembles J0.
"""

from __future__ import annotations

import numpy as np

from ._lib import SolverOpts
from .be_parallel import all_reduce_sum
from .fragsolver import default_opts
from .solver import ErrorMap


def calc_heff(fobj, pot, only_chem):
    """update_heff without side effects (numerical_jac.py:171-196)."""
    heff = np.zeros_like(fobj.h1)
    cout = fobj.udim
    edge_members = set()
    for e in fobj.relAO_per_edge:
        edge_members.update(e)
    for i in range(len(fobj.AO_in_frag)):
        if i not in edge_members:
            heff[i, i] -= pot[-1]
    if only_chem:
        return heff
    for e in fobj.relAO_per_edge:
        for j in range(len(e)):
            for k in range(j, len(e)):
                heff[e[j], e[k]] = pot[cout]
                heff[e[k], e[j]] = pot[cout]
                cout += 1
    return heff


class _Delta:
    """What ErrorMap.fill reads of a fragment, with the 1-RDM replaced."""

    def __init__(self, rdm1):
        self._rdm1 = rdm1


def compute_numerical_jacobian(beobj, solver="CCSD", only_chem=False, nproc=1, step_size=1e-6):
    if solver != "CCSD":
        raise ValueError("Solver not implemented")
    pot = np.asarray(beobj.pot if not only_chem else beobj.pot[-1:], dtype=float)
    npot = len(pot)
    J0 = np.zeros((npot, npot))
    # chemical potential: +-h sweeps over all fragments
    x = pot.copy(); x[-1] += step_size
    J0[:, -1] = beobj._sweep(list(x), only_chem=only_chem, eeval=False, return_vec=True)[1]
    x[-1] -= 2 * step_size
    J0[:, -1] -= beobj._sweep(list(x), only_chem=only_chem, eeval=False, return_vec=True)[1]
    J0[:, -1] /= 2 * step_size
    if only_chem:
        return J0
    # the amplitudes of the sweep just done stay on the device: every perturbed solve below starts from them
    opts = SolverOpts.from_buffer_copy(beobj.opts) if beobj.opts is not None else default_opts(beobj.lib)
    opts.warm_start = 1
    emap = beobj.emap if beobj.emap is not None else ErrorMap(beobj.Fobjs)
    nkpt = beobj.Fobjs[0].unitcell_nkpt
    cols = np.zeros((npot, npot))
    err = None
    try:
        for I in beobj.my_frags:
            f = beobj.Fobjs[I]
            for idx in range(f.udim, f.set_udim(f.udim)):
                rd = []
                for sgn in (+1.0, -1.0):
                    x = pot.copy(); x[idx] += sgn * step_size
                    out = f.dev.solve(f.nsocc, f.fock + calc_heff(f, x, only_chem), f.dm0, opts=opts, eeval=False)
                    rd.append(out["rdm1_emb"])
                view = [None] * len(beobj.Fobjs)
                view[I] = _Delta(rd[0] - rd[1])
                edge = np.zeros(emap.n_match); cen = np.zeros(emap.n_match)
                tr = emap.fill(view, [I], edge, cen) / nkpt
                cols[:, idx] = (np.append(edge, tr) - np.append(cen, 0.0)) / (2 * step_size)
    except Exception as e:  # noqa: BLE001 -- carried through the collective (be_parallel.all_reduce_sum)
        if beobj.world == 1:
            raise
        err = e
    if beobj.world > 1:
        all_reduce_sum(cols, error=err)
    J0[:, :-1] = cols[:, :-1]
    return J0
