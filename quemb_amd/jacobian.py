"""Initial Jacobian of the density-matching residual from fragment CPHF responses ('next' row f.1).

Mirror of `get_be_error_jacobian(n_frag, Fobjs, jac_solver="HF")` (shared/external/optqn.py:250-313) and
`get_atbe_Jblock_frag` (:316-392): for every fragment the RHF density response dP/d(lambda) to each unit matching
potential (`get_vpots_frag`, :464-490) and to the chemical potential comes from one device CPHF solve
(`qemb_frag_cphf`); the bookkeeping that scatters edge / centre elements into the block Jacobian is host logic.
jac_solver="HF" is the analytic one; the exact CCSD response of the sweep comes from `numerical_jac.compute_numerical_jacobian`
(`BE.optimize(jac_solver="Numerical")`).  The reference's "MP2" / "CCSD" options (jac_utils.py:162-178: an MP2-amplitude model of
the t1 response, built from per-perturbation integral derivatives on the host) are not reproduced.
"""

from __future__ import annotations

import numpy as np

from .be_parallel import all_reduce_sum


def get_vpots_frag(nao, relAO_per_edge, AO_in_frag):
    """Unit perturbations: one symmetric (j,k) pair per matching potential, then -1 on the non-edge fragment diagonal."""
    vp = []
    for e in relAO_per_edge:
        for j in range(len(e)):
            for k in range(j, len(e)):
                m = np.zeros((nao, nao))
                m[e[j], e[k]] = m[e[k], e[j]] = 1.0
                vp.append(m)
    m = np.zeros((nao, nao))
    edge_members = set(x for e in relAO_per_edge for x in e)
    for f in range(len(AO_in_frag)):
        if f not in edge_members:
            m[f, f] = -1.0
    vp.append(m)
    return vp


def _pairs(idx):
    return [(idx[j], idx[k]) for j in range(len(idx)) for k in range(j, len(idx))]


def jblock_frag(fobj, opts=None):
    """Per-fragment blocks (Je, Jc, xe, xc, y, alpha, ncout) as in get_atbe_Jblock_frag."""
    vpots = get_vpots_frag(fobj.nao, fobj.relAO_per_edge, fobj.AO_in_frag)
    dm0 = 2.0 * fobj._mo_coeffs[:, : fobj.nsocc] @ fobj._mo_coeffs[:, : fobj.nsocc].T
    dP = fobj.dev.cphf(fobj.nsocc, fobj.fock + fobj.heff, np.array(vpots), dm0=dm0, opts=opts)
    dPs, dP_mu = dP[:-1], dP[-1]
    edge_members = set(x for e in fobj.relAO_per_edge for x in e)
    nonedge = [f for f in range(len(fobj.AO_in_frag)) if f not in edge_members]
    edge_pairs = [pq for e in fobj.relAO_per_edge for pq in _pairs(e)]
    org_pairs = [(j, k) for j in fobj.relAO_per_origin for k in fobj.relAO_per_origin if j <= k]
    ncout = len(edge_pairs)
    Je = np.array([[dPs[c][p, q] for (p, q) in edge_pairs] for c in range(ncout)]).reshape(ncout, len(edge_pairs)).T
    Jc = np.array([[-dPs[c][p, q] for (p, q) in org_pairs] for c in range(ncout)]).reshape(ncout, len(org_pairs)).T
    y = [sum(dPs[c][f, f] for f in nonedge) for c in range(ncout)]
    xe = [dP_mu[p, q] for (p, q) in edge_pairs]
    xc = [-dP_mu[p, q] for (p, q) in org_pairs]
    alpha = sum(dP_mu[f, f] for f in nonedge)
    return Je, Jc, xe, xc, y, alpha, ncout


def get_be_error_jacobian(n_frag, Fobjs, jac_solver="HF", *, owner=None, rank=0, world=1, opts=None):
    if jac_solver.upper() != "HF":
        raise NotImplementedError("analytic Jacobian: jac_solver='HF' (CPHF on the device); use jac_solver='Numerical' for the CCSD response")
    # sizes are static; the CPHF blocks of the fragments this rank owns are computed here and summed over ranks
    ncouts = [sum(len(e) * (len(e) + 1) // 2 for e in f.relAO_per_edge) for f in Fobjs]
    norgs = [len([1 for j in f.relAO_per_origin for k in f.relAO_per_origin if j <= k]) for f in Fobjs]
    blocks = [None] * n_frag
    err = None
    try:
        for A in range(n_frag):
            if owner is None or owner[A] == rank:
                blocks[A] = jblock_frag(Fobjs[A], opts=opts)
    except Exception as e:  # noqa: BLE001 -- carried through the collective (be_parallel.all_reduce_sum)
        if world == 1:
            raise
        err = e
    if world > 1:
        # pack (Je, Jc, xe, xc, y, alpha) of every fragment into one buffer, zeros where not owned
        sizes = [nc * nc + no * nc + nc + no + nc + 1 for nc, no in zip(ncouts, norgs)]
        buf = np.zeros(sum(sizes))
        off = 0
        for A in range(n_frag):
            if blocks[A] is not None and err is None:
                Je, Jc, xe, xc, y, al, _ = blocks[A]
                buf[off: off + sizes[A]] = np.concatenate([Je.ravel(), Jc.ravel(), xe, xc, y, [al]])
            off += sizes[A]
        all_reduce_sum(buf, error=err)
        off = 0
        for A in range(n_frag):
            nc, no = ncouts[A], norgs[A]
            b = buf[off: off + sizes[A]]; off += sizes[A]
            p = 0
            Je = b[p: p + nc * nc].reshape(nc, nc); p += nc * nc
            Jc = b[p: p + no * nc].reshape(no, nc); p += no * nc
            xe = list(b[p: p + nc]); p += nc
            xc = list(b[p: p + no]); p += no
            y = list(b[p: p + nc]); p += nc
            blocks[A] = (Je, Jc, xe, xc, y, float(b[p]), nc)
    N_ = sum(ncouts)
    J = np.zeros((N_ + 1, N_ + 1))
    starts = np.concatenate([[0], np.cumsum(ncouts)])
    alpha = 0.0
    for A, fobj in enumerate(Fobjs):
        Je, Jc, xe, xc, y, al, nc = blocks[A]
        alpha += al
        s = starts[A]
        J[s: s + nc, s: s + nc] = Je
        J[s: s + nc, N_] = xe
        J[N_, s: s + nc] = y
        row = s
        for cidx in range(len(fobj.relAO_in_ref_per_edge)):
            ref = fobj.ref_frag_idx_per_edge[cidx]
            Jc_r, xc_r = blocks[ref][1], blocks[ref][3]
            nrow = Jc_r.shape[0]
            J[row: row + nrow, starts[ref]: starts[ref] + ncouts[ref]] += Jc_r
            J[row: row + nrow, N_] += xc_r
            row += nrow
    J[N_, N_] = alpha
    return J
