"""Periodic front-end of the hot path: the k -> R Fourier step and the SVD Schmidt decomposition of
kbe.pfrag.Frags.sd (kbe/pfrag.py:143-216), cons_h1 (:218-237) and get_nsocc (:264-306), with kbe.misc.get_phase /
get_phase1 (kbe/misc.py:24-34).

The supercell density `einsum("Rk,kuv,Sk->RuSv", phase, rdm1_lo_k, phase.conj())` (pfrag.py:176) and the back transform
`einsum("Rim,Rk->kim", TA_R, phase1)` (:192) are real GEMMs over stacked cos/sin factors on the device (FP64 MFMA); the
Schmidt step is `schmidt_decomp_svd` on the device (kbe/solver.py:9-46).  Per-k products of nlo x nlo complex matrices stay
on the host.  `cell` is replaced by its lattice vectors (`cell.lattice_vectors()`), the only thing the reference reads.
"""
from __future__ import annotations

import itertools

import numpy as np

from . import eri_transform as et
from .fragsolver import DeviceFragment
from .pfrag import Frags


def _Ts(kmesh):
    return np.array(list(itertools.product(range(kmesh[0]), range(kmesh[1]), range(kmesh[2]))), dtype=np.float64)


def get_phase(a_vec, kpts, kmesh):
    """kbe/misc.py:24-28."""
    Ts = _Ts(kmesh)
    return np.exp(1j * (Ts @ np.asarray(a_vec) @ np.asarray(kpts).T)) / np.sqrt(Ts.shape[0])


def get_phase1(a_vec, kpts, kmesh):
    """kbe/misc.py:31-34."""
    return np.exp(-1.0j * (_Ts(kmesh) @ np.asarray(a_vec) @ np.asarray(kpts).T))


class KFrags(Frags):
    """The periodic `Frags` state that sd / cons_h1 / get_nsocc read and write (kbe/pfrag.py:41-141)."""

    def __init__(self, AO_in_frag, weight_and_relAO_per_center=None, lib=None, ifrag=0, AO_per_edge=(), ref_frag_idx_per_edge=(),
                 relAO_per_edge=(), relAO_in_ref_per_edge=(), relAO_per_origin=None, unitcell=1, unitcell_nkpt=1.0):
        """kbe/pfrag.py:48-141.  Beyond the k-space pieces (sd, cons_h1, cons_fock, get_nsocc) a periodic fragment is the molecular
        one: scf / update_heff / set_udim / update_ebe_hf / solve are inherited from `pfrag.Frags` (the reference repeats them
        verbatim, kbe/pfrag.py:315-480)."""
        w = weight_and_relAO_per_center if weight_and_relAO_per_center is not None else (1.0, list(range(len(AO_in_frag))))
        super().__init__(AO_in_frag, ifrag, list(AO_per_edge), list(ref_frag_idx_per_edge), list(relAO_per_edge),
                         list(relAO_in_ref_per_edge), w, list(relAO_per_origin) if relAO_per_origin is not None else list(range(len(AO_in_frag))),
                         lib=lib)
        self.unitcell = unitcell
        self.unitcell_nkpt = unitcell_nkpt
        self.rdm1_lo_k = None

    def sd(self, lao, lmo, nocc, thr_bath=1.0e-10, a_vec=None, kpts=None, kmesh=None, h1=None):
        """kbe/pfrag.py:143-216.  Sets rdm1_lo_k, TA_lo_eo (nk, nlo, teo), TA (nk, nao, teo), nao = teo."""
        lao = np.asarray(lao); lmo = np.asarray(lmo)
        nk, nao, nlo = lao.shape
        rdm1_lo_k = np.stack([lmo[k][:, :nocc] @ lmo[k][:, :nocc].conj().T for k in range(nk)])
        self.rdm1_lo_k = rdm1_lo_k
        phase = get_phase(a_vec, kpts, kmesh)                       # (NR, nk)
        NR = phase.shape[0]
        # W[(R,S),k] = phase[R,k] conj(phase[S,k]);  D[(R,S),(u,v)] = sum_k W[(R,S),k] rdm[k,(u,v)]  (complex product as real GEMMs)
        W = (phase[:, None, :] * phase.conj()[None, :, :]).reshape(NR * NR, nk)
        Dk = rdm1_lo_k.reshape(nk, nlo * nlo)
        A_re = np.concatenate([W.real, -W.imag], axis=1)            # real part:  Wr Dr - Wi Di
        A_im = np.concatenate([W.imag, W.real], axis=1)             # imag part:  Wi Dr + Wr Di
        B = np.concatenate([Dk.real, Dk.imag], axis=0)
        sup_re = et.matmul(A_re, B, lib=self.lib)
        sup_im = et.matmul(A_im, B, lib=self.lib)
        if (max_val := np.abs(sup_im).max()) >= 1.0e-6:
            raise ValueError(f"Imaginary density in Full SD {max_val}")
        supcell_rdm = sup_re.reshape(NR, NR, nlo, nlo).transpose(0, 2, 1, 3).reshape(NR * nlo, NR * nlo)
        sites = [i + (nlo * 0) for i in self.AO_in_frag]
        TA_R = et.schmidt_decomp_svd(supcell_rdm, sites, thr_bath=thr_bath, lib=self.lib)
        teo = TA_R.shape[-1]
        # TA_k[k,(i,m)] = sum_R phase1[R,k] TA_R[R,(i,m)]
        phase1 = get_phase1(a_vec, kpts, kmesh)
        TAf = TA_R.reshape(NR, nlo * teo)
        TA_k = (et.matmul(phase1.real, TAf, transA=True, lib=self.lib) + 1j * et.matmul(phase1.imag, TAf, transA=True, lib=self.lib)).reshape(nk, nlo, teo)
        self.TA_lo_eo = TA_k
        self.TA = np.stack([lao[k] @ TA_k[k] for k in range(nk)])
        self.nao = self.TA.shape[-1]
        self.dev = DeviceFragment(self.nao, self.n_frag, lib=self.lib)
        return self.TA

    def real_space_TA(self, a_vec, kpts, kmesh):
        """The embedding orbitals in the AO basis of the Born-von-Karman supercell, rows ordered (cell R, AO mu) with the cells in the order
        of kbe.misc.get_phase: TA_R[(R, mu), i] = N_k^-1/2 sum_k phase[R, k] TA_k[mu, i] -- the inverse of the back transform of `sd`
        (kbe/pfrag.py:192).  Real for a time-reversal symmetric mean field (checked).  This is the rotation libdmet's
        `get_emb_eri_fast_gdf(cell, mydf, C_ao_eo=TA)` applies to the k-point GDF tensor (kbe/pbe.py:529-537)."""
        phase = get_phase(a_vec, kpts, kmesh)                        # (NR, nk)
        nk, nao, neo = self.TA.shape
        T = self.TA.reshape(nk, nao * neo)
        re = et.matmul(phase.real, T.real, lib=self.lib) - et.matmul(phase.imag, T.imag, lib=self.lib)
        im = et.matmul(phase.real, T.imag, lib=self.lib) + et.matmul(phase.imag, T.real, lib=self.lib)
        scale = 1.0 / np.sqrt(nk)
        if np.abs(im).max() * scale >= 1.0e-8:
            raise ValueError(f"Imaginary real-space embedding orbitals {np.abs(im).max() * scale}")
        return np.ascontiguousarray((re * scale).reshape(phase.shape[0] * nao, neo))

    def cons_h1(self, h1):
        """kbe/pfrag.py:218-237."""
        nk = self.TA.shape[0]
        h1_eo = sum(self.TA[k].conj().T @ h1[k] @ self.TA[k] for k in range(nk)) / float(nk)
        if np.abs(h1_eo.imag).max() < 1.0e-7:
            self.h1 = h1_eo.real
        else:
            raise ValueError(f"Imaginary Hcore {np.abs(h1_eo.imag).max()}")
        return self.h1

    def cons_fock(self, hf_veff, S, dm, eri_=None):
        """kbe/pfrag.py:240-268 with kbe/helper.py:11-60 (get_veff): the k-averaged projections of the density and of the mean field
        are host sums over k; J and K of the projected (real) density come from the device-resident fragment ERIs."""
        if eri_ is not None:
            self.set_eri(eri_)
        nk, nao, neo = self.TA.shape
        P_ = np.zeros((neo, neo), dtype=np.complex128)
        veff0 = np.zeros((neo, neo), dtype=np.complex128)
        for k in range(nk):
            Cinv = self.TA[k].conj().T @ S[k]
            P_ += Cinv @ dm[k] @ Cinv.conj().T
            veff0 += self.TA[k].conj().T @ hf_veff[k] @ self.TA[k]
        P_ = np.asarray((P_ / float(nk)).real, dtype=np.float64)
        veff0 /= float(nk)
        vj, vk = self.dev.jk(np.ascontiguousarray(P_))
        veff_ = veff0 - (vj - 0.5 * vk)
        if np.abs(veff_.imag).max() >= 1.0e-6:
            raise ValueError(f"Imaginary Veff {abs(veff_.imag).max()}")
        self.veff = veff_.real
        self.veff0 = veff0.real
        self.fock = self.h1 + veff_.real

    def get_nsocc(self, S, C, nocc, ncore=0):
        """kbe/pfrag.py:264-306: projected density, nsocc and the SVD guess orbitals (device SVD through nsocc_guess would need
        the MO factor; here P_ is formed per k on the host and diagonalised on the device)."""
        nk, nao, neo = self.TA.shape
        P_ = np.zeros((neo, neo), dtype=np.complex128)
        for k in range(nk):
            dm = 2.0 * (C[k][:, ncore: ncore + nocc] @ C[k][:, ncore: ncore + nocc].conj().T)
            Cinv = self.TA[k].conj().T @ S[k]
            P_ += Cinv @ dm @ Cinv.conj().T
        P_ /= float(nk)
        if np.abs(P_.imag).max() < 1.0e-6:
            P_ = P_.real
        else:
            raise ValueError(f"Imaginary density in get_nsocc {abs(P_.imag).max()}")
        self.nsocc = int(round(np.trace(P_).real) / 2)
        # svd(P_)[0] of the symmetric positive semi-definite P_ = its eigenvectors by descending eigenvalue
        w, V = et.eigh(P_, lib=self.lib)
        self._mo_coeffs = V[:, ::-1]
        return P_
