"""Initial Jacobian of the density-matching residual from fragment CPHF responses ('next' row f.1).

Mirror of `get_be_error_jacobian(n_frag, Fobjs, jac_solver="HF")` (shared/external/optqn.py:250-313) and
`get_atbe_Jblock_frag` (:316-392): for every fragment the RHF density response dP/d(lambda) to each unit matching
potential (`get_vpots_frag`, :464-490) and to the chemical potential comes from one device CPHF solve
(`qemb_frag_cphf`); the bookkeeping that scatters edge / centre elements into the block Jacobian is host logic.
jac_solver="HF" is the CPHF response; "MP2" and "CCSD" are the reference's correlated MODELS of the response (optqn.py:437-461):
"MP2" = derivative of the oo / vv blocks of the MP2 density (cpmp2_utils.py:94-133, halved), "CCSD" = twice the HF response plus the
derivative of an approximate t1 -- MP2 doubles put through one cycle of the CCSD t1 equation (jac_utils.py:13-41, :102-178).  Neither is
the exact CCSD response; that comes from `numerical_jac.compute_numerical_jacobian` (`BE.optimize(jac_solver="Numerical")`).
For the two models the device supplies the fragment RHF and every MO-integral block (`qemb_frag_prepare_ccsd` + block export: the
O(n^5) work); the per-perturbation algebra on those blocks -- O(o^2 v^3) per potential, the reference does an O(n^5) AO-basis
transformation per potential instead -- is host NumPy like the rest of the Jacobian bookkeeping.
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


class MoBlocks:
    """MO-integral blocks of one fragment exported from the device (chemists' notation, o = occupied, v = virtual):
    oooo[i,j,k,l] = (ij|kl), ovoo[i,a,j,k] = (ia|jk), ovov, oovv[i,j,a,b] = (ij|ab), ovvo[i,a,b,j] = (ia|bj), ovvv[i,a,b,c] = (ia|bc),
    vvvv[a,b,c,d] = (ab|cd); C, mo_energy of the fragment RHF they were transformed with."""

    def __init__(self, dev, nsocc, h, dm0, opts=None):
        n, o = dev.n, int(nsocc)
        v = n - o
        dev.prepare_ccsd(o, h, dm0, opts=opts)
        ex = dev.ccsd_export
        self.n, self.o, self.v = n, o, v
        self.C = ex("mo_coeff", (n, n))
        self.moe = np.concatenate([ex("eo", (o,)), ex("ev", (v,))])
        self.oooo, self.ovoo, self.ovov, self.ovvv = ex("oooo", (o, o, o, o)), ex("ovoo", (o, v, o, o)), ex("ovov", (o, v, o, v)), ex("ovvv", (o, v, v, v))
        self.ovvo = ex("W1base", (o, v, o, v)).transpose(2, 3, 1, 0)          # W1base[i,a,k,c] = ovvo[k,c,a,i]
        self.oovv = ex("W2base", (o, v, o, v)).transpose(2, 0, 1, 3)          # W2base[i,a,k,c] = oovv[k,i,a,c]
        self.vvvv = ex("Vl", (v, v, v, v)).transpose(0, 2, 1, 3)              # Vl[a,b,c,d] = (ac|bd)

    # ---- CPHF (cphf_utils.py:12-72): A = 4 (ia|jb) - (ib|ja) - (ij|ab) - diag(e_i - e_a); u = A^-1 B0, B0 = Co^T v Cv
    def cphf(self, vpots):
        o, v = self.o, self.v
        A = (4.0 * self.ovov - self.ovov.transpose(0, 3, 2, 1) - self.oovv.transpose(0, 2, 1, 3)).reshape(o * v, o * v)
        A = A - np.diag((self.moe[:o, None] - self.moe[None, o:]).ravel())
        B0 = np.stack([(self.C[:, :o].T @ q @ self.C[:, o:]).ravel() for q in vpots], axis=1)
        return np.linalg.solve(A, B0).T.reshape(len(vpots), o, v)

    # ---- dF = Q + J[2 dP] - K[2 dP]/2 with dP = -(Co u Cv^T + transpose) (cpmp2_utils.py:26-35), oo and vv blocks in the MO basis
    def dF_blocks(self, Q, u):
        o = self.o
        Co, Cv = self.C[:, :o], self.C[:, o:]
        es = lambda *a: np.einsum(*a, optimize=True)
        foo = Co.T @ Q @ Co - 4.0 * es("kcij,kc->ij", self.ovoo, u) + es("jcik,kc->ij", self.ovoo, u) + es("icjk,kc->ij", self.ovoo, u)
        fvv = Cv.T @ Q @ Cv - 4.0 * es("kcab,kc->ab", self.ovvv, u) + es("kabc,kc->ab", self.ovvv, u) + es("kbac,kc->ab", self.ovvv, u)
        return foo, fvv


def _t1_model(mo, Vovov, Voovo, Vvovv):
    """get_t1 (jac_utils.py:21-41): MP2-like doubles Vovov / D through the CCSD t1 equation.  Vvovv[c,j,b,a], Voovo[i,k,b,j]."""
    o = mo.o
    eia = mo.moe[:o, None] - mo.moe[None, o:]
    t2 = Vovov / (eia[:, :, None, None] + eia[None, None, :, :])
    es = lambda *a: np.einsum(*a, optimize=True)
    return (2.0 * es("ibjc,cjba->ia", t2, Vvovv) - es("jbic,cjba->ia", t2, Vvovv) - 2.0 * es("ikbj,jbka->ia", Voovo, t2)
            + es("ikbj,kbja->ia", Voovo, t2)) / eia


def dP_ccsd_model(mo, vpots):
    """get_dPccsdurlx_batch_u (jac_utils.py:162-178) with get_dt1ao_an (:102-159), in the MO basis of the exported blocks.
    A first-order orbital change d(occ i) = -sum_a u[i,a] |a>, d(virt a) = +sum_i u[i,a] |i> (get_dVmogen_r :59-66) turns every
    'one index transformed with dC' integral of the reference into a contraction of u with a neighbouring MO block."""
    o, v = mo.o, mo.v
    Co, Cv = mo.C[:, :o], mo.C[:, o:]
    es = lambda *a: np.einsum(*a, optimize=True)
    eia = mo.moe[:o, None] - mo.moe[None, o:]
    D2 = eia[:, :, None, None] + eia[None, None, :, :]
    Vovov = mo.ovov
    Vvovv = mo.ovvv.transpose(1, 0, 2, 3)                 # (cj|ba) = ovvv[j,c,b,a]
    Voovo = es("jbik->ikbj", mo.ovoo)                     # (ik|bj) = (jb|ik) = ovoo[j,b,i,k]
    t2 = Vovov / D2
    t1 = _t1_model(mo, Vovov, Voovo, Vvovv)
    us = mo.cphf(vpots)
    out = []
    for u, Q in zip(us, vpots):
        foo, fvv = mo.dF_blocks(Q, u)
        Aoo, Avv = -foo, -fvv
        tA = es("lajb,li->iajb", t2, Aoo) - es("idjb,da->iajb", t2, Avv)
        tA = tA + tA.transpose(2, 3, 0, 1)
        # dV blocks (get_dVmogen_r): sum over the four index positions of the block with that index rotated
        # dVovov[i,a,j,b] = -u[i,c] (ca|jb) + u[k,a] (ik|jb) + (ia <-> jb)
        h1 = -es("ic,jbca->iajb", u, mo.ovvv) + es("ka,jbik->iajb", u, mo.ovoo)
        dVovov = h1 + h1.transpose(2, 3, 0, 1)
        # dVvovv[c,j,b,a] = (cj|ba): c -> +u[k,c] (kj|ba); j -> -u[j,d] (cd|ba); b -> +u[k,b] (cj|ka); a -> +u[k,a] (cj|bk)
        dVvovv = (es("kc,kjba->cjba", u, mo.oovv) - es("jd,cdba->cjba", u, mo.vvvv) + es("kb,jcka->cjba", u, mo.ovov)
                  + es("ka,jcbk->cjba", u, mo.ovvo))
        # dVoovo[i,k,b,j] = (ik|bj): i -> -u[i,c] (ck|bj); k -> -u[k,c] (ic|bj); b -> +u[l,b] (ik|lj); j -> -u[j,c] (ik|bc)
        dVoovo = (-es("ic,kcbj->ikbj", u, mo.ovvo) - es("kc,icbj->ikbj", u, mo.ovvo) + es("lb,iklj->ikbj", u, mo.oooo)
                  - es("jc,ikbc->ikbj", u, mo.oovv))
        dt1 = (_t1_model(mo, tA, Voovo, Vvovv) + _t1_model(mo, dVovov, Voovo, Vvovv) + _t1_model(mo, Vovov, dVoovo, dVvovv)
               + (Aoo @ t1 - t1 @ Avv) / eia)
        dCo, dCv = -Cv @ u.T, Co @ u
        d = Co @ dt1 @ Cv.T + dCo @ t1 @ Cv.T + Co @ t1 @ dCv.T
        d = d + d.T
        hf = 2.0 * dCo @ Co.T
        out.append(d + hf + hf.T)
    return np.array(out)


def dP_mp2_model(mo, vpots):
    """mp2res_func (optqn.py:437-447) = get_dPmp2_batch_r (cpmp2_utils.py:94-133) / 2: derivative of the HF projector and of the oo / vv
    blocks of the MP2 density under the FULL orbital response U (get_full_u_F_r :42-59: the oo and vv rotations follow from dF)."""
    o, v, n = mo.o, mo.v, mo.n
    es = lambda *a: np.einsum(*a, optimize=True)
    eo, ev = mo.moe[:o], mo.moe[o:]
    eia = eo[:, None] - ev[None, :]
    D2 = eia[:, :, None, None] + eia[None, None, :, :]
    t2 = mo.ovov / D2

    def pmp2(tl, tr):           # get_Pmp2_r (:83-91)
        x = 2.0 * tr - tr.transpose(0, 3, 2, 1)
        P = np.zeros((n, n))
        P[:o, :o] = -es("iajb,majb->im", tl, x)
        P[o:, o:] = es("iajb,icjb->ac", tl, x)
        return P
    Phf = np.diag([1.0] * o + [0.0] * v)
    us = mo.cphf(vpots)
    out = []
    for u, Q in zip(us, vpots):
        foo, fvv = mo.dF_blocks(Q, u)
        dmoe = np.concatenate([np.diag(foo), np.diag(fvv)])                     # get_dmoe_F_r (:38-39)
        dD = (dmoe[:o, None] - dmoe[None, o:])
        dD2 = dD[:, :, None, None] + dD[None, None, :, :]
        Dij = -eo[:, None] + eo[None, :]; np.fill_diagonal(Dij, 1.0)
        Uoo = foo / Dij; np.fill_diagonal(Uoo, 0.0)
        Dab = -ev[:, None] + ev[None, :]; np.fill_diagonal(Dab, 1.0)
        Uvv = fvv / Dab; np.fill_diagonal(Uvv, 0.0)
        U = np.block([[Uoo, u], [-u.T, Uvv]])
        # get_dVovov_r (:62-80) with dC = C U: (ia|j dCv_b) = sum_p U[p,b] (ia|jp), (ia|dCo_j b) = sum_p U[p,j] (ia|pb)
        x1 = es("kb,iajk->iajb", U[:o, o:], mo.ovoo) + es("cb,iajc->iajb", U[o:, o:], mo.ovov)
        x2 = es("kj,iakb->iajb", U[:o, :o], mo.ovov) + es("cj,iacb->iajb", U[o:, :o], mo.ovvv)
        x = x1 + x2
        dVovov = x + x.transpose(2, 3, 0, 1)
        dt2 = (dVovov - t2 * dD2) / D2
        P = pmp2(t2, t2) + Phf
        dP = U @ P - P @ U
        dP2 = pmp2(dt2, t2)
        dP2 = dP2 + dP2.T
        out.append(0.5 * mo.C @ ((dP + dP2) * 2.0) @ mo.C.T)
    return np.array(out)


def jblock_frag(fobj, opts=None, jac_solver="HF"):
    """Per-fragment blocks (Je, Jc, xe, xc, y, alpha, ncout) as in get_atbe_Jblock_frag."""
    vpots = get_vpots_frag(fobj.nao, fobj.relAO_per_edge, fobj.AO_in_frag)
    dm0 = 2.0 * fobj._mo_coeffs[:, : fobj.nsocc] @ fobj._mo_coeffs[:, : fobj.nsocc].T
    if jac_solver == "HF":
        dP = fobj.dev.cphf(fobj.nsocc, fobj.fock + fobj.heff, np.array(vpots), dm0=dm0, opts=opts)
    else:
        mo = MoBlocks(fobj.dev, fobj.nsocc, fobj.fock + fobj.heff, dm0, opts=opts)
        dP = (dP_ccsd_model if jac_solver == "CCSD" else dP_mp2_model)(mo, np.array(vpots))
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
    jac_solver = jac_solver.upper()
    if jac_solver not in ("HF", "MP2", "CCSD"):
        raise NotImplementedError("Jacobian solver option not implemented.")      # optqn.py:264
    # sizes are static; the CPHF blocks of the fragments this rank owns are computed here and summed over ranks
    ncouts = [sum(len(e) * (len(e) + 1) // 2 for e in f.relAO_per_edge) for f in Fobjs]
    norgs = [len([1 for j in f.relAO_per_origin for k in f.relAO_per_origin if j <= k]) for f in Fobjs]
    blocks = [None] * n_frag
    err = None
    try:
        for A in range(n_frag):
            if owner is None or owner[A] == rank:
                blocks[A] = jblock_frag(Fobjs[A], opts=opts, jac_solver=jac_solver)
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
