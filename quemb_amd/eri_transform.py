"""AO -> fragment ERI transforms and Schmidt decomposition on the device (C ABI qemb_aoeri_*, qemb_df_*,
qemb_schmidt*).  Host mirror of the reference seams:

* `ao2mo.incore.full(eri_, TA, compact=True)`            molbe/mbe.py:1038            -> AOEri.transform
* `integral_direct_DF(mf, Fobjs, file_eri, auxbasis)`     molbe/eri_onthefly.py:45     -> DFContext.transform
* `transform_integral[_cuda](P_mu_nu, TA, S_abs, L_PQ, e)` molbe/eri_sparse_DF.py:677-702 -> DFContext (packed ints)
* `schmidt_decomposition(mo_coeff, nocc, AO_in_frag, ...)` molbe/pfrag.py:403           -> schmidt_decomposition
* `schmidt_decomp_svd(rdm, Frag_sites, thr_bath)`          kbe/solver.py:9              -> schmidt_decomp_svd
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import c_vp, check

#: new `int_transform` literals next to the reference's IntTransforms (molbe/mbe.py:63-71)
HIP_INT_TRANSFORMS = ("in-core-hip", "int-direct-DF-hip", "sparse-DF-hip", "on-fly-sparse-DF-hip")


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def matmul(A, B, transA=False, transB=False, lib=None):
    """op(A) @ op(B) on the device (FP64 MFMA GEMM), host arrays in and out."""
    lib = lib or _lib.init()
    A, B = _arr(A), _arr(B)
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != K2:
        raise ValueError(f"matmul: inner dimensions differ ({K} vs {K2})")
    out = np.empty((M, N))
    check(lib.qemb_matmul(M, N, K, A.ctypes.data, int(transA), B.ctypes.data, int(transB), out.ctypes.data), "qemb_matmul", lib)
    return out


class AOEri:
    """AO-basis ERIs resident on the device (uploaded once per system, shared by all fragments)."""

    def __init__(self, eri, nao: int, lib=None):
        self.lib = lib or _lib.init()
        eri = _arr(eri)
        npair = nao * (nao + 1) // 2
        if eri.ndim == 4 or eri.size == nao ** 4:
            sym = 1
        elif eri.size == npair * npair:
            sym = 4
        elif eri.size == npair * (npair + 1) // 2:
            sym = 8
        else:
            raise ValueError("AOEri: ERIs must be s1 (N^4), s4 (npair x npair) or s8 (1-D npair(npair))")
        self.nao = int(nao)
        h = c_vp()
        check(self.lib.qemb_aoeri_upload(self.nao, eri.ctypes.data, sym, C.byref(h)), "qemb_aoeri_upload", self.lib)
        self.h = h

    def transform(self, TA, frag=None, want_host=True):
        """(ij|kl) in the fragment embedding basis, 4-fold packed; optionally stored straight into `frag`."""
        TA = _arr(TA)
        if TA.shape[0] != self.nao:
            raise ValueError("AOEri.transform: TA has the wrong number of rows")
        n = TA.shape[1]
        npair = n * (n + 1) // 2
        out = np.empty((npair, npair)) if want_host else None
        check(self.lib.qemb_ao2mo_dense(self.h, TA.ctypes.data, n, None if out is None else out.ctypes.data,
                                        None if frag is None else frag.h), "qemb_ao2mo_dense", self.lib)
        return out

    def free(self):
        if getattr(self, "h", None):
            self.lib.qemb_aoeri_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ravel_symmetric(a: int, b: int) -> int:
    """_cpp/indexers.hpp:75-79."""
    return a * (a + 1) // 2 + b if a > b else b * (b + 1) // 2 + a


class SemiSparseSym3DTensor:
    """Host-side mirror of the reference's `SemiSparseSym3DTensor` (_cpp/eri_sparse_DF.cpp:110-298; Python binding :785-815):
    (P|mu nu) stored for the AO pairs of `exch_reachable` only, one aux vector per unique pair (nu <= mu), in the column order
    of `compute_offsets_and_unique` (:248-258).  `DFContext.set_ints_semisparse` uploads it -- or the reference's own object,
    which exposes the same `unique_dense_data` / `exch_reachable_with_offsets` -- without ever expanding it."""

    def __init__(self, shape, exch_reachable, unique_dense_data=None):
        naux, nao, nao2 = (int(x) for x in shape)
        if nao != nao2 or len(exch_reachable) != nao:
            raise ValueError("SemiSparseSym3DTensor: shape = (naux, nao, nao) and one reachable list per AO")       # :213-224
        self.shape = (naux, nao, nao)
        self.exch_reachable = [sorted(int(x) for x in r) for r in exch_reachable]
        self.exch_reachable_unique = [[nu for nu in r if nu <= mu] for mu, r in enumerate(self.exch_reachable)]   # indexers.hpp:149-162
        self.offsets = {}
        for mu, r in enumerate(self.exch_reachable_unique):
            for nu in r:
                self.offsets[ravel_symmetric(mu, nu)] = len(self.offsets)
        try:
            self.exch_reachable_with_offsets = [[(self.offsets[ravel_symmetric(mu, nu)], nu) for nu in r]
                                                for mu, r in enumerate(self.exch_reachable)]
        except KeyError:
            raise ValueError("SemiSparseSym3DTensor: exch_reachable must be symmetric (nu in reach(mu) <=> mu in reach(nu))") from None
        self.exch_reachable_unique_with_offsets = [[(o, nu) for o, nu in r if mu <= nu] for mu, r in enumerate(self.exch_reachable_with_offsets)]
        n_unique = len(self.offsets)
        if unique_dense_data is None:
            self.unique_dense_data = np.full((naux, n_unique), np.nan, order="F")                                  # :160
        else:
            self.unique_dense_data = np.asfortranarray(unique_dense_data, dtype=np.float64)
            if self.unique_dense_data.shape != (naux, n_unique):
                raise ValueError("SemiSparseSym3DTensor: unique_dense_data must be (naux, n_unique)")

    @property
    def mut_unique_dense_data(self):
        return self.unique_dense_data

    @property
    def nonzero_size(self):
        return self.unique_dense_data.size

    def get_aux_vector(self, mu, nu):
        return self.unique_dense_data[:, self.offsets[ravel_symmetric(mu, nu)]]

    @classmethod
    def from_dense(cls, P_mu_nu, exch_reachable):
        """Keep the stored pairs of a dense (naux, N, N) array."""
        P_mu_nu = np.asarray(P_mu_nu)
        t = cls(P_mu_nu.shape, exch_reachable)
        for mu, r in enumerate(t.exch_reachable_unique):
            for nu in r:
                t.unique_dense_data[:, t.offsets[ravel_symmetric(mu, nu)]] = P_mu_nu[:, mu, nu]
        return t


def _reach_csr(exch_reachable_with_offsets):
    ptr = np.zeros(len(exch_reachable_with_offsets) + 1, dtype=np.int64)
    for mu, r in enumerate(exch_reachable_with_offsets):
        ptr[mu + 1] = ptr[mu] + len(r)
    off = np.fromiter((o for r in exch_reachable_with_offsets for o, _ in r), dtype=np.int64, count=int(ptr[-1]))
    nu = np.fromiter((n for r in exch_reachable_with_offsets for _, n in r), dtype=np.int32, count=int(ptr[-1]))
    return ptr, nu, off


class DFContext:
    """Density-fitting context: metric factor and (P|mu nu) resident on the device."""

    def __init__(self, j2c=None, L_PQ=None, lib=None):
        self.lib = lib or _lib.init()
        h = c_vp()
        if (j2c is None) == (L_PQ is None):
            raise ValueError("DFContext: give exactly one of j2c (metric) or L_PQ (its lower Cholesky factor)")
        if j2c is not None:
            j2c = _arr(j2c)
            self.naux = j2c.shape[0]
            check(self.lib.qemb_df_create(self.naux, j2c.ctypes.data, C.byref(h)), "qemb_df_create", self.lib)
        else:
            L_PQ = _arr(L_PQ)
            self.naux = L_PQ.shape[0]
            check(self.lib.qemb_lpq_upload(L_PQ.ctypes.data, self.naux, C.byref(h)), "qemb_lpq_upload", self.lib)
        self.h = h
        self.nao = None

    @classmethod
    def periodic(cls, j2c, lib=None):
        """Context for the Gamma-point CC-GDF transform (kbe/eri_onthefly.py:160-164): the metric goes through `_j2c_cholesky_or_eig`
        (:19-45) on the device; `ischol` says which branch was taken."""
        self = cls.__new__(cls)
        self.lib = lib or _lib.init()
        j2c = _arr(j2c)
        self.naux = j2c.shape[0]
        h, ischol = c_vp(), C.c_int(-1)
        check(self.lib.qemb_df_create_pbc(self.naux, j2c.ctypes.data, C.byref(h), C.byref(ischol)), "qemb_df_create_pbc", self.lib)
        self.h, self.nao, self.ischol = h, None, bool(ischol.value)
        return self

    def alloc_ints(self, nao: int):
        """zeroed fitted tensor (L|mu nu) (real and imaginary part) for `add_pw_block` / `add_rs_block`"""
        check(self.lib.qemb_df_alloc_ints(self.h, int(nao)), "qemb_df_alloc_ints", self.lib)
        self.nao = int(nao)

    def add_pw_block(self, F, pw):
        """(L|mu nu) += sum_G F[L,G] (G|mu nu): F = ft_ao(chgcell, Gv_block).conj().T (naux, nG), pw = ft_aopair(cell, Gv_block) * coulG.conj()
        (nG, N, N), both complex   (kbe/eri_onthefly.py:176-199)"""
        F = np.asarray(F); pw = np.asarray(pw)
        nG = F.shape[1]
        if self.nao is None or F.shape[0] != self.naux or pw.shape != (nG, self.nao, self.nao):
            raise ValueError("DFContext.add_pw_block: alloc_ints first; F is (naux, nG) and pw (nG, N, N)")
        Fr, Fi, Pr, Pi = _arr(F.real), _arr(F.imag), _arr(pw.real), _arr(pw.imag)
        check(self.lib.qemb_df_add_pw_block(self.h, nG, Fr.ctypes.data, Fi.ctypes.data, Pr.ctypes.data, Pi.ctypes.data), "qemb_df_add_pw_block", self.lib)

    def add_rs_block(self, p0: int, block):
        """rows [p0, p0 + len(block)) of (L|mu nu) += block (real, (p1 - p0, N, N))   (kbe/eri_onthefly.py:201-217)"""
        block = _arr(block)
        if self.nao is None or block.shape[1:] != (self.nao, self.nao):
            raise ValueError("DFContext.add_rs_block: alloc_ints first; block is (p1 - p0, N, N)")
        check(self.lib.qemb_df_add_rs_block(self.h, int(p0), int(p0) + block.shape[0], block.ctypes.data), "qemb_df_add_rs_block", self.lib)

    def imag_absmax(self) -> float:
        out = C.c_double()
        check(self.lib.qemb_df_pw_imag_absmax(self.h, C.byref(out)), "qemb_df_pw_imag_absmax", self.lib)
        return out.value

    def select_part(self, part: int):
        check(self.lib.qemb_df_pw_select(self.h, int(part)), "qemb_df_pw_select", self.lib)

    def set_ints(self, ints, nao: int, layout: str = "pqL"):
        """layout: 'pqL' (N,N,naux) as getints3c returns, 'Lpq' (naux,N,N), 'packed' (naux, npair(N))."""
        code = {"pqL": 0, "Lpq": 1, "packed": 2}[layout]
        ints = _arr(ints)
        npair = nao * (nao + 1) // 2
        want = {0: nao * nao * self.naux, 1: nao * nao * self.naux, 2: self.naux * npair}[code]
        if ints.size != want:
            raise ValueError("DFContext.set_ints: array size does not match layout")
        check(self.lib.qemb_df_set_ints(self.h, int(nao), ints.ctypes.data, code), "qemb_df_set_ints", self.lib)
        self.nao = int(nao)

    def set_ints_semisparse(self, int_P_mu_nu):
        """Upload a SemiSparseSym3DTensor (this module's or the reference's pybind object: `unique_dense_data` (naux, n_unique) and
        `exch_reachable_with_offsets`) as it is; `transform` then runs transform_integral's algorithm on the sparse storage."""
        data = np.asarray(int_P_mu_nu.unique_dense_data)
        if data.ndim != 2 or data.shape[0] != self.naux:
            raise ValueError("DFContext.set_ints_semisparse: unique_dense_data must be (naux, n_unique)")
        rows = np.ascontiguousarray(data.T, dtype=np.float64)           # n_unique x naux; a view when the input is column-major
        if np.isnan(rows).any():
            raise ValueError("DFContext.set_ints_semisparse: unique_dense_data has unfilled (NaN) columns")       # eri_sparse_DF.py:494
        ptr, nu, off = _reach_csr(int_P_mu_nu.exch_reachable_with_offsets)
        nao = len(ptr) - 1
        check(self.lib.qemb_df_set_ints_semisparse(self.h, nao, rows.shape[0], rows.ctypes.data, ptr.ctypes.data, nu.ctypes.data,
                                                   off.ctypes.data), "qemb_df_set_ints_semisparse", self.lib)
        self.nao = nao

    def transform(self, TA, frag=None, want_host=True, S_abs=None, MO_coeff_epsilon=None, factor_only=False):
        """(ij|kl) 4-fold packed.  With `S_abs` and `MO_coeff_epsilon` the reference's semi-sparse screening is applied
        (transform_integral(int_P_mu_nu, TA, S_abs, L_PQ, MO_coeff_epsilon), molbe/eri_sparse_DF.py:677-678).
        factor_only (needs `frag`, no host copy): the fragment receives the fitted factor bb alone (eri_onthefly.py:141) and lives on it --
        the bb^T bb product of :143 is not formed (qemb_df_transform_factor); returns None."""
        TA = _arr(TA)
        if self.nao is None or TA.shape[0] != self.nao:
            raise ValueError("DFContext.transform: set_ints first / TA has the wrong number of rows")
        n = TA.shape[1]
        if factor_only:
            if frag is None or want_host:
                raise ValueError("DFContext.transform(factor_only=True) delivers into a fragment handle (frag=..., want_host=False)")
            if S_abs is not None:
                eps = 1e-5 if MO_coeff_epsilon is None else float(MO_coeff_epsilon)
                S_abs = _arr(S_abs)
                check(self.lib.qemb_df_transform_screened_factor(self.h, TA.ctypes.data, n, S_abs.ctypes.data, eps, frag.h),
                      "qemb_df_transform_screened_factor", self.lib)
            else:
                check(self.lib.qemb_df_transform_factor(self.h, TA.ctypes.data, n, frag.h), "qemb_df_transform_factor", self.lib)
            return None
        npair = n * (n + 1) // 2
        out = np.empty((npair, npair)) if want_host else None
        op = None if out is None else out.ctypes.data
        fh = None if frag is None else frag.h
        if S_abs is not None:
            S_abs = _arr(S_abs)
            eps = 1e-5 if MO_coeff_epsilon is None else float(MO_coeff_epsilon)     # reference default, mbe.py:189
            check(self.lib.qemb_df_transform_screened(self.h, TA.ctypes.data, n, S_abs.ctypes.data, eps, op, fh),
                  "qemb_df_transform_screened", self.lib)
        else:
            check(self.lib.qemb_df_transform(self.h, TA.ctypes.data, n, op, fh), "qemb_df_transform", self.lib)
        return out

    def free(self):
        if getattr(self, "h", None):
            self.lib.qemb_df_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def schmidt_decomposition(mo_coeff, nocc, AO_in_frag, thr_bath=1.0e-10, cinv=None, rdm=None, norb=None, lib=None, method="eigh"):
    """Same signature / return as molbe/pfrag.py:403-411: (TA_lo_eo, n_f, n_b).  The `cinv`, `rdm` and `norb`
    variants of the reference are only reached from UBE (out of scope) and raise here.
    method="eigh" is the reference's formulation (Jacobi eigh of the full environment block); method="subspace" gives the
    same bath through the rank-n_f invariant subspace of D[env,frag] (see csrc/schmidt.cpp) at a fraction of the cost."""
    if method not in ("eigh", "subspace"):
        raise ValueError("method must be 'eigh' or 'subspace'")
    if cinv is not None or rdm is not None or norb is not None:
        raise NotImplementedError("schmidt_decomposition: cinv / rdm / norb variants (UBE) are outside the hot path")
    lib = lib or _lib.init()
    C_ = _arr(mo_coeff)
    N, nmo = C_.shape
    frag = np.ascontiguousarray(AO_in_frag, dtype=np.int64)
    nf = len(frag)
    ld = min(N, 2 * nf + 2)
    while True:
        TA = np.empty((N, ld))
        nb, sw = C.c_int(), C.c_int()
        fn = lib.qemb_schmidt if method == "eigh" else lib.qemb_schmidt_subspace
        rc = fn(C_.ctypes.data, N, nmo, int(nocc), frag.ctypes.data_as(C.POINTER(C.c_int64)), nf, float(thr_bath),
                              TA.ctypes.data, ld, C.byref(nb), C.byref(sw))
        if rc == -1 and b"too narrow" in lib.qemb_last_error() and ld < N:
            ld = N          # more bath orbitals than fragment orbitals (non-idempotent input): retry with full width
            continue
        check(rc, "qemb_schmidt", lib)
        break
    return np.ascontiguousarray(TA[:, : nf + nb.value]), nf, nb.value


def schmidt_decomp_svd(rdm, Frag_sites, thr_bath=1.0e-10, lib=None):
    """kbe/solver.py:9-46 (returns the real TA; the reference's complex128 container holds real content)."""
    lib = lib or _lib.init()
    D = _arr(np.real(rdm))
    N = D.shape[0]
    frag = np.ascontiguousarray(Frag_sites, dtype=np.int64)
    nf = len(frag)
    TA = np.empty((N, 2 * nf))
    nb, sw = C.c_int(), C.c_int()
    check(lib.qemb_schmidt_svd(D.ctypes.data, N, frag.ctypes.data_as(C.POINTER(C.c_int64)), nf, float(thr_bath), TA.ctypes.data,
                               2 * nf, C.byref(nb), C.byref(sw)), "qemb_schmidt_svd", lib)
    return np.ascontiguousarray(TA[:, : nf + nb.value])


def eigh(A, lib=None):
    """Symmetric eigen-decomposition on the device (wavefront Jacobi): (w ascending, V columns)."""
    from ._lib import DeviceBuffer
    lib = lib or _lib.init()
    A = _arr(A)
    n = A.shape[0]
    dA = DeviceBuffer(n * n, lib=lib); dA.upload(A)
    dw, dV = DeviceBuffer(n, lib=lib), DeviceBuffer(n * n, lib=lib)
    check(lib.qemb_op_jacobi_eigh(n, dA.ptr, dw.ptr, dV.ptr, None), "qemb_op_jacobi_eigh", lib)
    w, V = dw.numpy((n,)), dV.numpy((n, n))
    for b in (dA, dw, dV):
        b.free()
    return w, V


def nsocc_guess(Cproj, lib=None):
    """Frags.get_nsocc core (molbe/pfrag.py:228-239): returns (P_, nsocc, mo_coeffs)."""
    lib = lib or _lib.init()
    Cp = _arr(Cproj)
    n, nocc = Cp.shape
    P = np.empty((n, n)); mo = np.empty((n, n))
    ns = C.c_int()
    check(lib.qemb_nsocc_guess(Cp.ctypes.data, n, nocc, P.ctypes.data, C.byref(ns), mo.ctypes.data), "qemb_nsocc_guess", lib)
    return P, ns.value, mo
