"""A periodic model with a consistent mean field, for the tests of the periodic driver (quemb_amd/kbe_pbe.py).

A ring of `nk` unit cells with `nlo` orthonormal orbitals each: translationally invariant one-body blocks h(0), h(+-1), and
density-fitted two-electron integrals (pq|rs) = sum_P B_P[p,q] B_P[r,s] whose auxiliary functions P = (cell R, a) are translated
copies of a local pattern over the cells R-1, R, R+1 -- 8-fold symmetric, positive, translationally invariant.  The closed-shell RHF
of the SUPERCELL (N = nk * nlo orbitals, the oracle's S = I solver on the dense tensor) gives a block-circulant Fock matrix; its
Fourier blocks F_k = sum_D f(D) exp(-i k D a) are diagonalised per k-point -- the same mean field seen by a k-point code.  Two views
of one system come out:

    supercell view   a molecular-like mean field (N x N real matrices, dense ERIs) + one BE2 fragment per site (N fragments)
    k-point view     `KMeanField` (per-k hcore, overlap, orbitals, mean-field potential, energy per cell) + one BE2 fragment per site
                     of the reference cell (nlo fragments, edges matched to translated copies), fragment ERIs from the supercell
                     tensor rotated with the real-space image of TA_k

Per unit cell both must give the same HF-in-HF identity and the same correlation energy (tests/test_kbe_pbe.py).
Test infrastructure (uses oracle/)."""
import numpy as np

from qemb_oracle import eri as oeri
from qemb_oracle import scf as oscf


def build(nk=4, nlo=3, nocc_cell=1, seed=5, naux_cell=4, gap=1.6, scale=0.45):
    rng = np.random.default_rng(seed)
    N = nk * nlo
    a = 2.5
    h0 = rng.standard_normal((nlo, nlo)); h0 = 0.15 * (h0 + h0.T) + np.diag(gap * np.arange(nlo))
    h1 = 0.25 * rng.standard_normal((nlo, nlo))                      # block between cell R and cell R+1
    hs = np.zeros((N, N))
    for R in range(nk):
        s0 = slice(R * nlo, (R + 1) * nlo)
        s1 = slice(((R + 1) % nk) * nlo, ((R + 1) % nk + 1) * nlo)
        hs[s0, s0] += h0
        hs[s0, s1] += h1
        hs[s1, s0] += h1.T
    # local auxiliary pattern over three neighbouring cells (3 nlo orbitals), symmetric in its two orbital indices
    pat = scale * rng.standard_normal((naux_cell, 3 * nlo, 3 * nlo))
    pat = 0.5 * (pat + pat.transpose(0, 2, 1))
    decay = np.repeat([0.5, 1.0, 0.5], nlo)
    pat = pat * decay[None, :, None] * decay[None, None, :]
    B = np.zeros((nk * naux_cell, N, N))
    for R in range(nk):
        idx = np.concatenate([np.arange(((R + d) % nk) * nlo, ((R + d) % nk + 1) * nlo) for d in (-1, 0, 1)])
        for a_ in range(naux_cell):
            B[R * naux_cell + a_][np.ix_(idx, idx)] += pat[a_]
    eri = np.einsum("Ppq,Prs->pqrs", B, B, optimize=True)
    nocc = nk * nocc_cell
    mf = oscf.rhf(hs, eri, nocc, conv_tol=1e-13, conv_tol_grad=1e-10)
    assert mf["converged"]
    e_gap = mf["mo_energy"][nocc] - mf["mo_energy"][nocc - 1]
    assert e_gap > 0.3, e_gap
    dm = mf["dm"]
    vj, vk = oscf.get_jk(eri, dm)
    veff = vj - 0.5 * vk
    F = hs + veff
    # block-circulant check and Fourier blocks:  A_k = sum_D a(D) exp(-i k D a),  a(D) = block (cell D, cell 0)
    blk = lambda A, R, S: A[R * nlo:(R + 1) * nlo, S * nlo:(S + 1) * nlo]
    for R in range(nk):
        assert np.abs(blk(F, R, (R + 1) % nk) - blk(F, 0, 1)).max() < 1e-8, "the supercell RHF broke the translational symmetry"
    kpts = np.array([[2 * np.pi * m / (nk * a), 0.0, 0.0] for m in range(nk)])
    a_vec = np.diag([a, 12.0, 12.0])
    ft = lambda A: np.stack([sum(blk(A, D, 0) * np.exp(-1j * kpts[m][0] * D * a) for D in range(nk)) for m in range(nk)])
    hk, Fk, vk_ = ft(hs), ft(F), ft(veff)
    Ck, ek = [], []
    for m in range(nk):
        e, V = np.linalg.eigh(Fk[m])
        Ck.append(V); ek.append(e)
    ek = np.array(ek)
    assert np.sort(ek[:, :nocc_cell].ravel()).max() < np.sort(ek[:, nocc_cell:].ravel()).min(), "aufbau is not uniform over k"
    return dict(nk=nk, nlo=nlo, N=N, a=a, a_vec=a_vec, kpts=kpts, kmesh=[nk, 1, 1], nocc_cell=nocc_cell, nocc=nocc,
                h_super=hs, eri_super=eri, B=B, mf_super=mf, veff_super=veff, F_super=F,
                hk=hk, Sk=np.stack([np.eye(nlo, dtype=np.complex128)] * nk), Ck=np.array(Ck), ek=ek, veffk=vk_,
                e_tot_cell=mf["e_tot"] / nk)


def ring_be2_lists(nsite, ncentres):
    """BE2 fragments of a ring of `nsite` one-orbital sites, one per centre 0..ncentres-1: fragment c = {c, c-1, c+1}; the edge c+-1 is
    matched to the centre of fragment (c+-1) mod ncentres -- for ncentres < nsite (the reference cell of a periodic system) that is a
    translated copy of a fragment of the set."""
    AO = [[c, (c - 1) % nsite, (c + 1) % nsite] for c in range(ncentres)]
    edges = [[[(c - 1) % nsite], [(c + 1) % nsite]] for c in range(ncentres)]
    ref = [[(c - 1) % ncentres, (c + 1) % ncentres] for c in range(ncentres)]
    rel_edge = [[[1], [2]] for _ in range(ncentres)]
    rel_ref = [[[0], [0]] for _ in range(ncentres)]
    return dict(AO_per_frag=AO, AO_per_edge_per_frag=edges, ref_frag_idx_per_edge_per_frag=ref,
                relAO_per_origin_per_frag=[[0] for _ in range(ncentres)],
                weight_and_relAO_per_center_per_frag=[(1.0, [0]) for _ in range(ncentres)],
                relAO_per_edge_per_frag=rel_edge, relAO_in_ref_per_edge_per_frag=rel_ref, n_BE=2)


class SupercellMF:
    """the attributes quemb_amd.mbe.BE reads from a molecular mean-field object"""

    def __init__(self, m):
        self.m = m
        self.mol = type("Mol", (), {"nelectron": 2 * m["nocc"]})()
        self.mo_coeff, self.mo_energy, self.e_tot = m["mf_super"]["mo_coeff"], m["mf_super"]["mo_energy"], m["mf_super"]["e_tot"]
        self._eri = oeri.pack_s4(m["eri_super"])

    def energy_nuc(self): return 0.0
    def get_hcore(self): return self.m["h_super"]
    def get_ovlp(self): return np.eye(self.m["N"])
    def make_rdm1(self): return self.m["mf_super"]["dm"]
    def get_veff(self, dm=None): return self.m["veff_super"]


def real_space_TA(f, m):
    """TA_R[(R,mu), i] = (1/nk) sum_k exp(+i k R a) TA_k[mu, i]: the inverse of KFrags.sd's back transform (kbe/pfrag.py:192)."""
    nk, nlo, a = m["nk"], m["nlo"], m["a"]
    TA = np.zeros((nk, nlo, f.TA.shape[-1]), dtype=np.complex128)
    for R in range(nk):
        TA[R] = sum(np.exp(1j * m["kpts"][k][0] * R * a) * f.TA[k] for k in range(nk)) / nk
    assert np.abs(TA.imag).max() < 1e-9
    return np.ascontiguousarray(TA.real.reshape(nk * nlo, -1))
