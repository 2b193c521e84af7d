"""Frags -- host mirror of the reference's fragment object (molbe/pfrag.py:38-400) over the device solver.

Same attribute names and method meanings as the reference so that be_func / BEOPT read the same; the
numerics behind `sd`, `cons_fock`, `get_nsocc`, `scf` and the correlated solve run in libqemb_hip.
Differences, all deliberate (DESIGN.md "Boundary"):
* fragment ERIs live in HBM inside `self.dev` (a DeviceFragment) instead of dataset `f{I}` of eri_file.h5;
* `scf()` without `fs=True` only records the inputs: the fragment RHF of a sweep runs inside the device
  solve together with CCSD (one C-ABI call per fragment per sweep);
* no PySCF objects (`_mf`, `_mc`): `mo_coeffs`, `mo_energy`, `t1`, `_rdm1`, `rdm1__` are plain arrays.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np

from . import eri_transform as et
from .fragsolver import DeviceFragment, default_opts


class Frags:
    def __init__(self, AO_in_frag: Sequence[int], ifrag: int, AO_per_edge, ref_frag_idx_per_edge, relAO_per_edge,
                 relAO_in_ref_per_edge, weight_and_relAO_per_center, relAO_per_origin, eri_file=None,
                 unrestricted: bool = False, lib=None):
        if unrestricted:
            raise NotImplementedError("unrestricted fragments (UBE) are outside the accelerated hot path")
        self.AO_in_frag = list(AO_in_frag)
        self.n_frag = len(self.AO_in_frag)
        self.AO_per_edge = AO_per_edge
        self.ref_frag_idx_per_edge = ref_frag_idx_per_edge
        self.relAO_per_edge = relAO_per_edge
        self.relAO_in_ref_per_edge = relAO_in_ref_per_edge
        self.relAO_per_origin = relAO_per_origin
        self.weight_and_relAO_per_center = (float(weight_and_relAO_per_center[0]), list(weight_and_relAO_per_center[1]))
        self.eri_file = eri_file
        self.ifrag = ifrag
        self.dname = "f" + str(ifrag)
        self.lib = lib
        self.dev: DeviceFragment | None = None
        self.TA = None
        self.TA_lo_eo = None
        self.n_f = self.n_b = 0
        self.h1 = None
        self.nao = 0
        self.mo_coeffs = None
        self._mo_coeffs = None
        self.mo_energy = None
        self.nsocc = 0
        self.t1 = None
        self.t2 = None
        self.heff = None
        self.udim = None
        self._rdm1 = None
        self.rdm1__ = None
        self.rdm2__ = None
        self.ebe = 0.0
        self.ebe_hf = 0.0
        self.fock = None
        self.veff = None
        self.veff0 = None
        self.dm0 = None
        self.unitcell_nkpt = 1.0
        self._hf_jk = None

    # ------------------------------------------------------------------ Schmidt (pfrag.py:146-180)
    def sd(self, lao, lmo, nocc, thr_bath, norb=None, method="eigh"):
        if norb is not None:
            raise NotImplementedError("norb (UBE) is outside the hot path")
        self.TA_lo_eo, self.n_f, self.n_b = et.schmidt_decomposition(lmo, nocc, self.AO_in_frag, thr_bath=thr_bath, lib=self.lib,
                                                                      method=method)
        self.TA = et.matmul(lao, self.TA_lo_eo, lib=self.lib)
        self.nao = self.TA.shape[1]
        self.dev = DeviceFragment(self.nao, self.n_frag, lib=self.lib)

    def set_eri(self, eri_s4):
        """Store 4-fold packed fragment ERIs on the device (what dataset f{I} holds in the reference)."""
        if self.dev is None:
            self.nao = int(round((np.sqrt(8 * eri_s4.shape[0] + 1) - 1) / 2))
            self.dev = DeviceFragment(self.nao, self.n_frag, lib=self.lib)
        self.dev.set_eri_s4(eri_s4)

    # ------------------------------------------------------------------ Fock (pfrag.py:182-206, helper.py:28-69)
    def cons_fock(self, hf_veff, S, dm, eri_=None):
        if eri_ is not None:
            self.set_eri(eri_)
        ST = S @ self.TA
        P_ = ST.T @ dm @ ST
        vj, vk = self.dev.jk(np.asarray(P_.real, dtype=np.float64))
        veff0 = self.TA.T @ hf_veff @ self.TA
        self.veff = veff0 - (vj - 0.5 * vk)
        self.veff0 = veff0
        self.fock = self.h1 + self.veff

    # ------------------------------------------------------------------ nsocc (pfrag.py:208-239)
    def get_nsocc(self, S, C, nocc, ncore=0):
        C_ = self.TA.T @ S @ C[:, ncore:ncore + nocc]
        P_, self.nsocc, self._mo_coeffs = et.nsocc_guess(C_, lib=self.lib)
        return P_

    # ------------------------------------------------------------------ SCF (pfrag.py:241-288)
    def scf(self, heff=None, fs=False, eri=None, dm0=None, opts=None):
        if eri is not None:
            self.set_eri(eri)
        if heff is None:
            heff = self.heff
        if dm0 is None:
            dm0 = 2.0 * self._mo_coeffs[:, : self.nsocc] @ self._mo_coeffs[:, : self.nsocc].T
        r = self.dev.scf(self.nsocc, self.fock + heff, dm0, opts=opts)
        if not r["converged"]:
            raise RuntimeError(f"fragment {self.ifrag}: SCF did not converge")
        if fs:
            self._mo_coeffs = r["mo_coeff"].copy()
            self._hf_jk = (r["J"], r["K"])
        else:
            self.mo_coeffs = r["mo_coeff"].copy()
        self.mo_energy = r["mo_energy"]
        return r

    # ------------------------------------------------------------------ potentials (pfrag.py:290-325)
    def update_heff(self, u, cout=None, only_chem=False):
        heff_ = np.zeros_like(self.h1)
        if cout is None:
            cout = self.udim
        edge_members = set()
        for e in self.relAO_per_edge:
            edge_members.update(e)
        for i in range(self.n_frag):
            if i not in edge_members:
                heff_[i, i] -= u[-1]
        if not only_chem:
            for e in self.relAO_per_edge:
                ne = len(e)
                for j in range(ne):
                    for k in range(j, ne):
                        heff_[e[j], e[k]] = u[cout]
                        heff_[e[k], e[j]] = u[cout]
                        cout += 1
        self.heff = heff_

    def set_udim(self, cout):
        for e in self.relAO_per_edge:
            cout += len(e) * (len(e) + 1) // 2
        return cout

    # ------------------------------------------------------------------ HF energy (pfrag.py:327-400)
    def update_ebe_hf(self, rdm_hf=None, mo_coeffs=None, eri=None, return_e=False):
        """e_i = 2 h1.D + veff.D + sum_j D_ij (2 J_ij - K_ij), i < n_frag, D = C_o C_o^T; weighted centre sum.
        The J/K of D come from the device (the reference loops over packed ERI rows, pfrag.py:365-383)."""
        if mo_coeffs is None:
            mo_coeffs = self._mo_coeffs
        if rdm_hf is None:
            rdm_hf = mo_coeffs[:, : self.nsocc] @ mo_coeffs[:, : self.nsocc].T
        nf = self.n_frag
        e1 = 2.0 * np.einsum("ij,ij->i", self.h1[:nf], rdm_hf[:nf])
        ec = np.einsum("ij,ij->i", self.veff[:nf], rdm_hf[:nf])
        J, K = self.dev.jk(rdm_hf)
        e2 = np.einsum("ij,ij->i", rdm_hf[:nf], (2.0 * J - K)[:nf])
        e_ = e1 + e2 + ec
        w, cen = self.weight_and_relAO_per_center
        self.ebe_hf = float(sum(w * e_[i] for i in cen))
        if return_e:
            return (sum(w * e1[i] for i in cen), sum(w * (e2[i] + ec[i]) for i in cen), e_)
        return None

    # ------------------------------------------------------------------ the sweep body
    def solve(self, opts=None, eeval=True, use_cumulant=True, want_t2=False, relax_density=False):
        """update_heff -> scf -> solve_ccsd -> rdm1 -> get_frag_energy for this fragment (solver.py:301-547).
        relax_density: solve_ccsd(relax=True) (solver.py:925-939) -- Lambda equations on the device, response densities."""
        opts = self._solve_inputs(opts, eeval, relax_density)
        out = self.dev.solve(self.nsocc, self.fock + self.heff, self.dm0, opts=opts, eeval=eeval, want_t2=want_t2)
        return self._solve_outputs(out, eeval, use_cumulant)

    def _solve_inputs(self, opts, eeval, relax_density):
        """what has to be on the device before the solve call (shared with the lock-step sweep, solver.solve_fragments)"""
        if bool(relax_density) != bool(getattr(opts, "relax_density", 0) if opts is not None else 0):
            from ._lib import SolverOpts
            from .fragsolver import default_opts
            opts = SolverOpts.from_buffer_copy(opts) if opts is not None else default_opts(self.dev.lib)
            opts.relax_density = int(bool(relax_density))
        if eeval:
            w, cen = self.weight_and_relAO_per_center
            self.dev.set_energy_data(self.h1, self.veff0, self.veff, w, cen)
        return opts

    def _solve_outputs(self, out, eeval, use_cumulant):
        """what solver.py:493-505 sets on the fragment object"""
        self.mo_coeffs = out["mo_coeff"]
        self.mo_energy = out["mo_energy"]
        self.t1 = out["t1"]
        self.t2 = out["t2"]
        self.rdm1__ = out["rdm1_mo"]
        self._rdm1 = out["rdm1_emb"]
        if eeval and not use_cumulant:
            out["e_frag"] = self._noncumulant_energy(out)
        return out

    def _noncumulant_energy(self, out):
        """get_frag_energy(use_cumulant=False) (helper.py:292-339): the 2-RDM then carries the mean-field pieces
        (make_rdm2_urlx(with_dm1=True), ccsd_rdm.py:40-53).  They are bilinear in D0 = C_o C_o^T and the first-order
        change D' = C [[0,t1],[t1^T,0]] C^T, so their contraction with the fragment ERIs reduces to J/K builds on the
        device:  e2_P += sum_Q D0_PQ (J[D'] - K[D']/2 + 2 J[D0] - K[D0])_PQ + D'_PQ (J[D0] - K[D0]/2)_PQ."""
        nf, o = self.n_frag, self.nsocc
        C = out["mo_coeff"]
        D0 = C[:, :o] @ C[:, :o].T
        rdm = out["rdm1_emb"]
        Dp = 2.0 * (rdm - D0)
        J0, K0 = self.dev.jk(D0)
        Jp, Kp = self.dev.jk(Dp)
        e2x = np.einsum("ij,ij->i", D0[:nf], (Jp - 0.5 * Kp + 2.0 * J0 - K0)[:nf]) + np.einsum("ij,ij->i", Dp[:nf], (J0 - 0.5 * K0)[:nf])
        e1 = 2.0 * np.einsum("ij,ij->i", self.h1[:nf], rdm[:nf])
        ec = np.einsum("ij,ij->i", self.veff[:nf], rdm[:nf])
        w, cen = self.weight_and_relAO_per_center
        e2c = out["e_frag"][1]                       # cumulant part, already centre-summed on the device
        return np.array([w * sum(e1[c] for c in cen), e2c + w * sum(e2x[c] for c in cen), w * sum(ec[c] for c in cen)])
