"""Oracle for rows a10-a15: fragment energies, heff scatter, density-matching residual and the fragment
sweep (be_func).  Test infrastructure."""
from dataclasses import dataclass, field

import numpy as np

from . import ccsd as occsd
from . import rdm as ordm
from . import scf as oscf
from .eri import restore_s1, pack_s4


@dataclass
class Frag:
    """The subset of molbe/pfrag.py:38 `Frags` state the hot path reads/writes."""
    AO_in_frag: list
    ifrag: int
    AO_per_edge: list
    ref_frag_idx_per_edge: list
    relAO_per_edge: list
    relAO_in_ref_per_edge: list
    weight_and_relAO_per_center: tuple
    relAO_per_origin: list = field(default_factory=list)
    TA: np.ndarray = None
    n_f: int = 0
    n_b: int = 0
    nao: int = 0
    nsocc: int = 0
    h1: np.ndarray = None
    fock: np.ndarray = None
    veff: np.ndarray = None
    veff0: np.ndarray = None
    heff: np.ndarray = None
    eri_s4: np.ndarray = None
    dm0: np.ndarray = None
    _mo_coeffs: np.ndarray = None
    mo_coeffs: np.ndarray = None
    mo_energy: np.ndarray = None
    t1: np.ndarray = None
    t2: np.ndarray = None
    _rdm1: np.ndarray = None
    rdm1__: np.ndarray = None
    udim: int = 0
    ebe_hf: float = 0.0
    unitcell_nkpt: float = 1.0

    @property
    def n_frag(self):
        return len(self.AO_in_frag)


def initialize_pot(n_frag, relAO_per_edge_per_frag):
    """molbe/mbe.py:1614-1650."""
    pot = []
    if relAO_per_edge_per_frag:
        for I in range(n_frag):
            for e in relAO_per_edge_per_frag[I]:
                for j in range(len(e)):
                    for k in range(len(e)):
                        if j > k:
                            continue
                        pot.append(0.0)
    pot.append(0.0)
    return pot


def set_udim(fr, cout):
    """molbe/pfrag.py:318-325."""
    for e in fr.relAO_per_edge:
        n = len(e)
        cout += n * (n + 1) // 2
    return cout


def update_heff(fr, u, cout=None, only_chem=False):
    """molbe/pfrag.py:290-316: -u[-1] on non-edge diagonal (:297-299); edge blocks in upper-tri order (:305-314)."""
    heff = np.zeros_like(fr.h1)
    if cout is None:
        cout = fr.udim
    for i in range(len(fr.AO_in_frag)):
        if not any(i in sub for sub in fr.relAO_per_edge):
            heff[i, i] -= u[-1]
    if not only_chem:
        for e in fr.relAO_per_edge:
            for j in range(len(e)):
                for k in range(len(e)):
                    if j > k:
                        continue
                    heff[e[j], e[k]] = u[cout]
                    heff[e[k], e[j]] = u[cout]
                    cout += 1
    fr.heff = heff
    return heff


def update_ebe_hf(fr, mo_coeffs=None, return_e=False):
    """molbe/pfrag.py:327-400 (restricted): fragment HF energy with D = C_o C_o^T."""
    if mo_coeffs is None:
        mo_coeffs = fr._mo_coeffs
    nf = fr.n_frag
    rdm_hf = mo_coeffs[:, :fr.nsocc] @ mo_coeffs[:, :fr.nsocc].T
    e1 = 2.0 * np.einsum("ij,ij->i", fr.h1[:nf], rdm_hf[:nf])
    ec = np.einsum("ij,ij->i", fr.veff[:nf], rdm_hf[:nf])
    jmax = fr.TA.shape[1]
    eri = fr.eri_s4
    e2 = np.zeros_like(e1)
    il = np.tril_indices(jmax)
    for i in range(nf):
        for j in range(jmax):
            ij = i * (i + 1) // 2 + j if i > j else j * (j + 1) // 2 + i
            Gij = (2.0 * rdm_hf[i, j] * rdm_hf - np.outer(rdm_hf[i], rdm_hf[j]))[:jmax, :jmax]
            Gij[np.diag_indices(jmax)] *= 0.5
            Gij += Gij.T
            e2[i] += Gij[il] @ eri[ij]
    e_ = e1 + e2 + ec
    w, cen = fr.weight_and_relAO_per_center
    fr.ebe_hf = sum(w * e_[i] for i in cen)
    if return_e:
        return (sum(w * e1[i] for i in cen), sum(w * (e2[i] + ec[i]) for i in cen), e_)
    return fr.ebe_hf


def get_frag_energy(mo_coeffs, nsocc, n_frag, weight_and_relAO_per_center, TA, h1, rdm1, rdm2s, eri_s4, veff0,
                    veff=None, use_cumulant=True):
    """molbe/helper.py:220-339.  `eri_s4` replaces the h5 read at :303-304."""
    rdm1s_rot = mo_coeffs @ rdm1 @ mo_coeffs.T * 0.5
    hf_1rdm = mo_coeffs[:, :nsocc] @ mo_coeffs[:, :nsocc].T
    if use_cumulant:
        delta = 2 * (rdm1s_rot - hf_1rdm)
        e1 = np.einsum("ij,ij->i", h1[:n_frag], delta[:n_frag])
        ec = np.einsum("ij,ij->i", veff0[:n_frag], delta[:n_frag])
    else:
        e1 = 2 * np.einsum("ij,ij->i", h1[:n_frag], rdm1s_rot[:n_frag])
        ec = np.einsum("ij,ij->i", veff[:n_frag], rdm1s_rot[:n_frag])
    jmax = TA.shape[1]
    r2 = np.einsum("ijkl,pi,qj,rk,sl->pqrs", 0.5 * rdm2s, *([mo_coeffs] * 4), optimize=True)
    e2 = np.zeros_like(e1)
    il = np.tril_indices(jmax)
    for i in range(n_frag):
        for j in range(jmax):
            ij = i * (i + 1) // 2 + j if i > j else j * (j + 1) // 2 + i
            Gij = r2[i, j, :jmax, :jmax].copy()
            Gij[np.diag_indices(jmax)] *= 0.5
            Gij += Gij.T
            e2[i] += Gij[il] @ eri_s4[ij]
    w, cen = weight_and_relAO_per_center
    return [sum(w * e1[i] for i in cen), sum(w * e2[i] for i in cen), sum(w * ec[i] for i in cen)]


def solve_error(Fobjs, Nocc, only_chem=False):
    """molbe/solver.py:683-778."""
    err_edge = []
    err_chempot = 0.0
    if only_chem:
        for f in Fobjs:
            for i in f.weight_and_relAO_per_center[1]:
                err_chempot += f._rdm1[i, i]
        err_chempot /= Fobjs[0].unitcell_nkpt
        err = err_chempot - Nocc
        return abs(err), np.asarray([err])
    for f in Fobjs:
        for edge in f.relAO_per_edge:
            for j in range(len(edge)):
                for k in range(len(edge)):
                    if j > k:
                        continue
                    err_edge.append(f._rdm1[edge[j], edge[k]])
        for i in f.weight_and_relAO_per_center[1]:
            err_chempot += f._rdm1[i, i]
    err_chempot /= Fobjs[0].unitcell_nkpt
    err_edge.append(err_chempot)
    err_cen = []
    for f in Fobjs:
        for cidx, cens in enumerate(f.relAO_in_ref_per_edge):
            for j in range(len(cens)):
                for k in range(len(cens)):
                    if j > k:
                        continue
                    err_cen.append(Fobjs[f.ref_frag_idx_per_edge[cidx]]._rdm1[cens[j], cens[k]])
    err_cen.append(Nocc)
    err_vec = np.array(err_edge) - np.array(err_cen)
    return float(np.mean(err_vec * err_vec) ** 0.5), err_vec


def solve_fragment(fr, use_cumulant=True, eeval=True, ccsd_kw=None, relax_density=False):
    """One pass of the body of be_func's loop (molbe/solver.py:301-547) == run_solver (be_parallel.py:40-307):
    fragment RHF -> CCSD -> rdm1 back-rotation -> fragment energy."""
    ccsd_kw = ccsd_kw or {}
    n = fr.TA.shape[1]
    dm0 = fr.dm0 if fr.dm0 is not None else 2.0 * fr._mo_coeffs[:, :fr.nsocc] @ fr._mo_coeffs[:, :fr.nsocc].T
    mf = oscf.rhf(fr.fock + fr.heff, fr.eri_s4, fr.nsocc, dm0=dm0)
    fr.mo_coeffs = mf["mo_coeff"].copy(); fr.mo_energy = mf["mo_energy"].copy()
    t1, t2, ecorr_mo, nit = occsd.solve_ccsd(None, fr.eri_s4, fr.nsocc, mf["mo_coeff"], mf["mo_energy"], **ccsd_kw)
    fr.t1, fr.t2 = t1, t2
    rdm1 = ordm.make_rdm1_ccsd_t1(t1)
    if relax_density:        # solve_ccsd(relax=True), molbe/solver.py:925-939
        from . import ccsd_lambda as olam
        eris = occsd.Eris(fr.eri_s4, mf["mo_coeff"], fr.nsocc, mo_energy=mf["mo_energy"])
        z1, z2, _, lag = olam.solve_lambda(t1, t2, eris, conv_tol=1e-11)
        rdm1, _ = olam.response_densities(lag, z1, z2)
    fr.rdm1__ = rdm1.copy()
    fr._rdm1 = fr.mo_coeffs @ rdm1 @ fr.mo_coeffs.T * 0.5
    e_f = None
    if eeval:
        if relax_density:
            rdm2 = olam.make_rdm2_relaxed(lag, z1, z2)
            if not use_cumulant:
                rdm2 = ordm.add_dm1_terms(rdm2, rdm1, fr.nsocc)
        else:
            rdm2 = ordm.make_rdm2_urlx(t1, t2, with_dm1=not use_cumulant)
        e_f = get_frag_energy(fr.mo_coeffs, fr.nsocc, fr.n_frag, fr.weight_and_relAO_per_center, fr.TA, fr.h1, rdm1,
                              rdm2, fr.eri_s4, fr.veff0, fr.veff, use_cumulant)
    return e_f, nit, ecorr_mo


def be_func(pot, Fobjs, Nocc, only_chem=False, eeval=False, return_vec=False, use_cumulant=True, ccsd_kw=None, relax_density=False):
    """molbe/solver.py:244-562 restricted to solver == 'CCSD'."""
    total_e = [0.0, 0.0, 0.0]
    for f in Fobjs:
        if pot is not None:
            update_heff(f, pot, only_chem=only_chem)
        e_f, _, _ = solve_fragment(f, use_cumulant=use_cumulant, eeval=eeval, ccsd_kw=ccsd_kw, relax_density=relax_density)
        if eeval:
            total_e = [a + b for a, b in zip(total_e, e_f)]
            update_ebe_hf(f)
    Ecorr = sum(total_e)
    if eeval and not return_vec:
        return Ecorr, total_e
    ernorm, ervec = solve_error(Fobjs, Nocc, only_chem=only_chem)
    if eeval:
        return ernorm, ervec, [Ecorr, total_e]
    return (ernorm, ervec, None) if return_vec else ernorm


def init_fragment(fr, W, lmo_coeff, Nocc, hcore, S, C, hf_dm, hf_veff, eri_ao, thr_bath=1e-10):
    """What BE.initialize does to one fragment (mbe.py:1205-1229, :1116-1174): Schmidt, ERI transform,
    nsocc, h1, cons_fock, fragment SCF (fs=True), dm0, ebe_hf."""
    from .schmidt import schmidt_decomposition, get_nsocc
    from .eri import ao2mo_full
    TA_lo, nf, nb = schmidt_decomposition(lmo_coeff, Nocc, fr.AO_in_frag, thr_bath=thr_bath)
    fr.TA = W @ TA_lo; fr.n_f, fr.n_b = nf, nb; fr.nao = fr.TA.shape[1]
    fr.eri_s4 = ao2mo_full(eri_ao, fr.TA, compact=True)
    _, fr.nsocc, fr._mo_coeffs = get_nsocc(fr.TA, S, C, Nocc)
    fr.h1 = fr.TA.T @ hcore @ fr.TA
    fr.veff, fr.veff0 = oscf.get_veff(fr.eri_s4, hf_dm, S, fr.TA, hf_veff)
    fr.fock = fr.h1 + fr.veff
    fr.heff = np.zeros_like(fr.h1)
    dm0 = 2.0 * fr._mo_coeffs[:, :fr.nsocc] @ fr._mo_coeffs[:, :fr.nsocc].T
    mf = oscf.rhf(fr.fock + fr.heff, fr.eri_s4, fr.nsocc, dm0=dm0)
    fr._mo_coeffs = mf["mo_coeff"].copy()
    fr.dm0 = 2.0 * fr._mo_coeffs[:, :fr.nsocc] @ fr._mo_coeffs[:, :fr.nsocc].T
    update_ebe_hf(fr)
    return fr
