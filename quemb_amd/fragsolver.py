"""Python handle on one device-resident fragment (C ABI qemb_frag_* in include/qemb_hip.h).

This is the object `quemb_amd.pfrag.Frags` keeps in place of the reference's HDF5 dataset name + PySCF
objects: fragment ERIs stay in HBM between sweeps (the reference re-reads `eri_file.h5` every call --
molbe/helper.py:182-189, :303-304) and one `solve()` is the body of be_func's loop (molbe/solver.py:301-547).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SolverOpts, c_vp, check


def _p(a):
    return None if a is None else a.ctypes.data


def default_opts(lib=None, **kw) -> SolverOpts:
    lib = lib or _lib.init()
    o = SolverOpts()
    lib.qemb_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


class DeviceFragment:
    def __init__(self, n: int, n_f: int, lib=None):
        self.lib = lib or _lib.init()
        self.n, self.n_f = int(n), int(n_f)
        h = c_vp()
        check(self.lib.qemb_frag_create(self.n, self.n_f, C.byref(h)), "qemb_frag_create", self.lib)
        self.h = h

    # ---- data --------------------------------------------------------------------------------
    def set_eri_s4(self, eri_s4: np.ndarray):
        npair = self.n * (self.n + 1) // 2
        a = np.ascontiguousarray(eri_s4, dtype=np.float64)
        if a.shape != (npair, npair):
            raise ValueError(f"fragment ERIs must be 4-fold packed ({npair},{npair}), got {a.shape}")
        check(self.lib.qemb_frag_set_eri_s4(self.h, a.ctypes.data), "qemb_frag_set_eri_s4", self.lib)

    def set_eri_s4_dev(self, dev_ptr: int):
        check(self.lib.qemb_frag_set_eri_s4_dev(self.h, dev_ptr), "qemb_frag_set_eri_s4_dev", self.lib)

    def set_df_factor(self, B: np.ndarray):
        """the fitted 3-index factor B (naux, npair(n)) with eri_s4 = B.T @ B (`bb` of molbe/eri_onthefly.py:141-143): solves then form their MO integrals
        from it while that is the cheaper route (qemb_frag_mo_route).  Call after set_eri_s4 (new ERIs drop the factor)."""
        npair = self.n * (self.n + 1) // 2
        a = np.ascontiguousarray(B, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != npair:
            raise ValueError(f"the 3-index factor must be (naux, {npair}), got {a.shape}")
        check(self.lib.qemb_frag_set_df_factor(self.h, int(a.shape[0]), a.ctypes.data), "qemb_frag_set_df_factor", self.lib)

    def set_df_factor_dev(self, dev_ptr: int, naux: int):
        check(self.lib.qemb_frag_set_df_factor_dev(self.h, int(naux), dev_ptr), "qemb_frag_set_df_factor_dev", self.lib)

    def set_df_only(self, B: np.ndarray):
        """the fragment lives on its 3-index factor B (naux, npair(n)) alone -- no 4-fold packed block resident (qemb_frag_set_df_only): J / K, MO integrals,
        energies and responses come from B; `get_eri_s4` forms B.T @ B on demand"""
        npair = self.n * (self.n + 1) // 2
        a = np.ascontiguousarray(B, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != npair:
            raise ValueError(f"the 3-index factor must be (naux, {npair}), got {a.shape}")
        check(self.lib.qemb_frag_set_df_only(self.h, int(a.shape[0]), a.ctypes.data), "qemb_frag_set_df_only", self.lib)

    def set_df_only_dev(self, dev_ptr: int, naux: int):
        check(self.lib.qemb_frag_set_df_only_dev(self.h, int(naux), dev_ptr), "qemb_frag_set_df_only_dev", self.lib)

    def resident_bytes(self) -> int:
        b = C.c_int64()
        check(self.lib.qemb_frag_resident_bytes(self.h, C.byref(b)), "qemb_frag_resident_bytes", self.lib)
        return int(b.value)

    def set_mo_route(self, route: int):
        """-1: the cheaper route (default), 0: four-index transformation of the packed block, 1: the 3-index factor"""
        check(self.lib.qemb_frag_mo_route(self.h, int(route)), "qemb_frag_mo_route", self.lib)

    def mo_route_used(self):
        """(the last solve formed its MO integrals from the factor, naux of the factor held -- 0: none)"""
        used, naux = C.c_int(), C.c_int()
        check(self.lib.qemb_frag_mo_route_used(self.h, C.byref(used), C.byref(naux)), "qemb_frag_mo_route_used", self.lib)
        return bool(used.value), int(naux.value)

    def get_eri_s4(self) -> np.ndarray:
        npair = self.n * (self.n + 1) // 2
        out = np.empty((npair, npair))
        check(self.lib.qemb_frag_get_eri_s4(self.h, out.ctypes.data), "qemb_frag_get_eri_s4", self.lib)
        return out

    def set_energy_data(self, h1, veff0, veff, weight, centers):
        h1 = np.ascontiguousarray(h1, dtype=np.float64)
        veff0 = np.ascontiguousarray(veff0, dtype=np.float64)
        veff = None if veff is None else np.ascontiguousarray(veff, dtype=np.float64)
        cen = np.ascontiguousarray(centers, dtype=np.int32)
        check(self.lib.qemb_frag_set_energy_data(self.h, h1.ctypes.data, veff0.ctypes.data, _p(veff), float(weight),
                                                 cen.ctypes.data_as(C.POINTER(C.c_int)), len(cen)), "qemb_frag_set_energy_data", self.lib)

    def jk(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        J = np.empty((self.n, self.n)); K = np.empty((self.n, self.n))
        check(self.lib.qemb_frag_jk(self.h, P.ctypes.data, J.ctypes.data, K.ctypes.data), "qemb_frag_jk", self.lib)
        return J, K

    # ---- the sweep body -------------------------------------------------------------------------
    def solve(self, nsocc, h, dm0=None, opts: SolverOpts | None = None, eeval=True, want_t2=False):
        n, o = self.n, int(nsocc)
        v = n - o
        h = np.ascontiguousarray(h, dtype=np.float64)
        dm0 = None if dm0 is None else np.ascontiguousarray(dm0, dtype=np.float64)
        opts = opts or default_opts(self.lib)
        out = dict(mo_coeff=np.empty((n, n)), mo_energy=np.empty(n), rdm1_emb=np.empty((n, n)), rdm1_mo=np.empty((n, n)),
                   t1=np.empty((o, v)), t2=np.empty((o, o, v, v)) if want_t2 else None)
        e_frag = np.zeros(3)
        ecorr, escf, ebehf = C.c_double(), C.c_double(), C.c_double()
        nit, ncyc = C.c_int(), C.c_int()
        check(self.lib.qemb_frag_solve(self.h, o, h.ctypes.data, _p(dm0), C.byref(opts), int(bool(eeval)),
                                       out["mo_coeff"].ctypes.data, out["mo_energy"].ctypes.data, out["rdm1_emb"].ctypes.data,
                                       out["rdm1_mo"].ctypes.data, out["t1"].ctypes.data, _p(out["t2"]), e_frag.ctypes.data,
                                       C.byref(ecorr), C.byref(escf), C.byref(ebehf), C.byref(nit), C.byref(ncyc)),
              "qemb_frag_solve", self.lib)
        nlam = C.c_int()
        check(self.lib.qemb_frag_lambda_iters(self.h, C.byref(nlam)), "qemb_frag_lambda_iters", self.lib)
        out.update(e_frag=e_frag, e_corr_mo=ecorr.value, e_scf=escf.value, ebe_hf=ebehf.value, n_iter=nit.value,
                   scf_cycles=ncyc.value, lambda_iters=nlam.value)
        return out

    def scf(self, nsocc, h, dm0=None, opts=None):
        """Fragment RHF only (Frags.scf(fs=True)); returns dict(mo_coeff, mo_energy, J, K, e_scf, converged, cycles)."""
        n = self.n
        h = np.ascontiguousarray(h, dtype=np.float64)
        dm0 = None if dm0 is None else np.ascontiguousarray(dm0, dtype=np.float64)
        opts = opts or default_opts(self.lib)
        mo = np.empty((n, n)); eps = np.empty(n); J = np.empty((n, n)); K = np.empty((n, n))
        e = C.c_double(); conv = C.c_int(); cyc = C.c_int()
        check(self.lib.qemb_frag_scf(self.h, int(nsocc), h.ctypes.data, _p(dm0), C.byref(opts), mo.ctypes.data, eps.ctypes.data,
                                     J.ctypes.data, K.ctypes.data, C.byref(e), C.byref(conv), C.byref(cyc)), "qemb_frag_scf", self.lib)
        return dict(mo_coeff=mo, mo_energy=eps, J=J, K=K, e_scf=e.value, converged=bool(conv.value), cycles=cyc.value)

    def cphf(self, nsocc, h, vpots, dm0=None, opts=None):
        """Density responses dP_p (npot, n, n) to the one-body perturbations vpots (npot, n, n)."""
        n = self.n
        h = np.ascontiguousarray(h, dtype=np.float64)
        v = np.ascontiguousarray(vpots, dtype=np.float64).reshape(-1, n, n)
        dm0 = None if dm0 is None else np.ascontiguousarray(dm0, dtype=np.float64)
        opts = opts or default_opts(self.lib)
        out = np.empty_like(v)
        check(self.lib.qemb_frag_cphf(self.h, int(nsocc), h.ctypes.data, _p(dm0), C.byref(opts), v.ctypes.data, v.shape[0],
                                      out.ctypes.data), "qemb_frag_cphf", self.lib)
        return out

    # ---- measurement hooks ---------------------------------------------------------------------------
    def prepare_ccsd(self, nsocc, h, dm0=None, opts=None):
        h = np.ascontiguousarray(h, dtype=np.float64)
        dm0 = None if dm0 is None else np.ascontiguousarray(dm0, dtype=np.float64)
        opts = opts or default_opts(self.lib)
        check(self.lib.qemb_frag_prepare_ccsd(self.h, int(nsocc), h.ctypes.data, _p(dm0), C.byref(opts)), "qemb_frag_prepare_ccsd", self.lib)

    def ccsd_iterate(self, niter=1):
        e, nt = C.c_double(), C.c_double()
        check(self.lib.qemb_frag_ccsd_iterate(self.h, int(niter), C.byref(e), C.byref(nt)), "qemb_frag_ccsd_iterate", self.lib)
        return e.value, nt.value

    def ccsd_export(self, name: str, shape):
        out = np.empty(shape)
        check(self.lib.qemb_frag_ccsd_export(self.h, name.encode(), out.ctypes.data, out.size), "qemb_frag_ccsd_export", self.lib)
        return out

    def ccsd_reset(self):
        check(self.lib.qemb_frag_ccsd_reset(self.h), "qemb_frag_ccsd_reset", self.lib)

    def free(self):
        if getattr(self, "h", None):
            self.lib.qemb_frag_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def solve_batch(frags, nsoccs, hs, dm0s=None, opts: SolverOpts | None = None, eeval=True, want_t2=False, stats=None):
    """qemb_frag_solve_batch: every fragment of `frags` (DeviceFragment objects of one library) in one call -- fragment RHF, MO transformation
    and the density / energy evaluation per fragment on its own stream, the CCSD iterations of all fragments in lock step (one grouped
    launch per operation).  Returns the list of dicts DeviceFragment.solve would return, bit for bit; `stats` (a dict) receives the
    launch counters of the lock-step iterations."""
    F = len(frags)
    if F == 0:
        return []
    lib = frags[0].lib
    opts = opts or default_opts(lib)
    ns = [int(o) for o in nsoccs]
    hs = [np.ascontiguousarray(h, dtype=np.float64) for h in hs]
    dm0s = [None] * F if dm0s is None else [None if d is None else np.ascontiguousarray(d, dtype=np.float64) for d in dm0s]
    outs = []
    for fr, o in zip(frags, ns):
        n, v = fr.n, fr.n - o
        outs.append(dict(mo_coeff=np.empty((n, n)), mo_energy=np.empty(n), rdm1_emb=np.empty((n, n)), rdm1_mo=np.empty((n, n)),
                         t1=np.empty((o, v)), t2=np.empty((o, o, v, v)) if want_t2 else None))
    VP = C.c_void_p * F
    arr = lambda xs: VP(*[None if x is None else x.ctypes.data for x in xs])
    handles = VP(*[fr.h for fr in frags])
    nso = (C.c_int * F)(*ns)
    e_frag = np.zeros((F, 3)); ecorr = np.zeros(F); escf = np.zeros(F); ebehf = np.zeros(F)
    nit = (C.c_int * F)(); ncyc = (C.c_int * F)(); st = (C.c_int64 * 5)()
    check(lib.qemb_frag_solve_batch(F, handles, nso, arr(hs), arr(dm0s), C.byref(opts), int(bool(eeval)),
                                    arr([o_["mo_coeff"] for o_ in outs]), arr([o_["mo_energy"] for o_ in outs]), arr([o_["rdm1_emb"] for o_ in outs]),
                                    arr([o_["rdm1_mo"] for o_ in outs]), arr([o_["t1"] for o_ in outs]), arr([o_["t2"] for o_ in outs]),
                                    e_frag.ctypes.data, ecorr.ctypes.data, escf.ctypes.data, ebehf.ctypes.data, nit, ncyc, st),
          "qemb_frag_solve_batch", lib)
    for f, (fr, o_) in enumerate(zip(frags, outs)):
        nlam = C.c_int()
        check(lib.qemb_frag_lambda_iters(fr.h, C.byref(nlam)), "qemb_frag_lambda_iters", lib)
        o_.update(e_frag=e_frag[f].copy(), e_corr_mo=float(ecorr[f]), e_scf=float(escf[f]), ebe_hf=float(ebehf[f]), n_iter=int(nit[f]),
                  scf_cycles=int(ncyc[f]), lambda_iters=nlam.value)
    if stats is not None:
        for k, name in enumerate(("merged_runs", "launches", "grouped_launches", "operations", "max_group")):
            stats[name] = stats.get(name, 0) + int(st[k]) if name != "max_group" else max(stats.get(name, 0), int(st[k]))
    return outs
