"""be_func / solve_error / solve_ccsd -- host mirror of molbe/solver.py over the device fragment solver.

`be_func` keeps the reference's signature (molbe/solver.py:244-257) for solver == "CCSD"; every other solver
string of the reference (MP2, FCI, SCI, DMRG, ...) is a different code path of QuEmb that this package does
not replace and raises ValueError("Solver not implemented") exactly like the reference's final else (:490-491).
"""

from __future__ import annotations

import numpy as np

from ._lib import SolverOpts
from .fragsolver import DeviceFragment, default_opts


class ErrorMap:
    """Static index map of solve_error (molbe/solver.py:683-778): which (fragment, row, col) of the fragment 1-RDMs
    feed each slot of err_edge / err_cen.  Built once; evaluating the residual is then two gathers -- and in the
    multi-GPU sweep the same map tells each rank which slots of the all-reduce buffer it owns."""

    def __init__(self, Fobjs):
        ef, er, ec = [], [], []
        cf, cr, cc = [], [], []
        cenf, ceni = [], []
        for fidx, f in enumerate(Fobjs):
            for edge in f.relAO_per_edge:
                for j in range(len(edge)):
                    for k in range(j, len(edge)):
                        ef.append(fidx); er.append(edge[j]); ec.append(edge[k])
            for i in f.weight_and_relAO_per_center[1]:
                cenf.append(fidx); ceni.append(i)
        for f in Fobjs:
            for cidx, cens in enumerate(f.relAO_in_ref_per_edge):
                ref = f.ref_frag_idx_per_edge[cidx]
                for j in range(len(cens)):
                    for k in range(j, len(cens)):
                        cf.append(ref); cr.append(cens[j]); cc.append(cens[k])
        self.edge = (np.array(ef, dtype=int), np.array(er, dtype=int), np.array(ec, dtype=int))
        self.cen = (np.array(cf, dtype=int), np.array(cr, dtype=int), np.array(cc, dtype=int))
        self.diag = (np.array(cenf, dtype=int), np.array(ceni, dtype=int))
        if len(self.edge[0]) != len(self.cen[0]):
            raise ValueError("edge / centre matching lists have different lengths")
        self.n_match = len(self.edge[0])

    def fill(self, Fobjs, owned, edge_vals, cen_vals):
        """Add the contributions of the fragments in `owned` to the (zero-initialised) buffers; returns the
        centre-diagonal partial sum (the chemical-potential condition)."""
        owned = set(owned)
        tr = 0.0
        f, r, c = self.edge
        for s in range(self.n_match):
            if f[s] in owned:
                edge_vals[s] += Fobjs[f[s]]._rdm1[r[s], c[s]]
        f, r, c = self.cen
        for s in range(self.n_match):
            if f[s] in owned:
                cen_vals[s] += Fobjs[f[s]]._rdm1[r[s], c[s]]
        f, i = self.diag
        for s in range(len(f)):
            if f[s] in owned:
                tr += Fobjs[f[s]]._rdm1[i[s], i[s]]
        return tr


def solve_error(Fobjs, Nocc, only_chem=False, rdm1_list=None):
    """molbe/solver.py:683-778: (norm, err_vec).  err_vec = [edge elements ..., sum centre diag / nkpt] -
    [centre elements read from the reference fragments ..., Nocc]; norm = sqrt(mean(err^2))."""
    if rdm1_list is not None:
        class _V:  # lightweight view so that ErrorMap can index `._rdm1`
            def __init__(self, f, r):
                self.__dict__.update(f.__dict__); self._rdm1 = r
        Fobjs = [_V(f, r) for f, r in zip(Fobjs, rdm1_list)]
    emap = ErrorMap(Fobjs)
    edge = np.zeros(emap.n_match); cen = np.zeros(emap.n_match)
    tr = emap.fill(Fobjs, range(len(Fobjs)), edge, cen) / Fobjs[0].unitcell_nkpt
    if only_chem:
        err = tr - Nocc
        return abs(err), np.asarray([err])
    err_vec = np.append(edge, tr) - np.append(cen, Nocc)
    return float(np.mean(err_vec * err_vec) ** 0.5), err_vec


def solve_ccsd(h, eri_s4, nsocc, dm0=None, *, n_frag=0, rdm_return=False, rdm2_return=False, relax=False, use_cumulant=True,
               opts=None, lib=None):
    """Device counterpart of solve_ccsd (molbe/solver.py:829-946).  The reference takes a PySCF mean-field object;
    here the fragment RHF is part of the device call, so the inputs are what `get_scfObj` would have been given:
    h = fock + heff, the 4-fold packed fragment ERIs, nsocc and dm0.
    Returns (t1, t2) or (t1, t2, rdm1_mo, mo_coeff) with rdm_return.  The dense 2-RDM is never formed on the device
    (rdm2_return raises): its only consumer, get_frag_energy, is evaluated in contracted form by `Frags.solve`."""
    if relax:
        opts = SolverOpts.from_buffer_copy(opts) if opts is not None else default_opts(lib)
        opts.relax_density = 1
    if rdm2_return:
        raise NotImplementedError("the n^4 2-RDM is not materialised; use Frags.solve(eeval=True) for energies")
    n = h.shape[0]
    fr = DeviceFragment(n, n_frag, lib=lib)
    fr.set_eri_s4(eri_s4)
    out = fr.solve(nsocc, h, dm0, opts=opts, eeval=False, want_t2=True)
    fr.free()
    if rdm_return:
        return out["t1"], out["t2"], out["rdm1_mo"], out["mo_coeff"]
    return out["t1"], out["t2"]


def fragment_work_bytes(n, o=None):
    """Device memory ONE fragment in flight takes beside its resident ERIs: the two n^2 x npair buffers of the
    embedding->MO transformation (rows at a stride of whole 128-byte lines), the (+/-) pair-packed ladder operands, the ovvv block with its
    packed images, and ~30 tensors of the size of t2.  Without n_occ the worst split (n_occ = n / 4) is assumed."""
    n = int(n)
    o = max(1, n // 4) if o is None else int(o)
    v = max(n - o, 1)
    ld = (n + 15) // 16 * 16
    npair = n * (n + 1) // 2
    pv, qv = v * (v + 1) // 2, v * (v - 1) // 2
    return 8.0 * (2.0 * n * ld * npair + pv * pv + qv * qv + 2.0 * o * v ** 3 + 30.0 * (o * v) ** 2)


def sweep_mode(frags, nstreams=None, lockstep=None, mem_free=None):
    """How the fragments of a sweep share the GPU when the caller leaves it open (BE(..., nstreams=None, lockstep=None)):
    many small fragments (>= 4 of at most 64 embedding orbitals: launch bound) advance in lock step -- one grouped launch per
    operation for all of them;
    otherwise several fragments are in flight on separate streams: up to six small ones, up to four of at most 256 orbitals, two beyond --
    and never more than the device memory that is free (`mem_free` bytes; taken from the fragments' library when they are on a device)
    holds working sets of (fragment_work_bytes): a calculation that fits one fragment at a time keeps fitting with the default.
    Measured on octane/STO-3G: BE2 (six fragments of ~42 orbitals) 40 ms serial, 18.3 ms six streams, 16.7 ms lock step; BE3 (four of ~55) 41 / 28 / 29 ms.
    Every mode returns bit-identical results."""
    frags = list(frags)
    nmax = max((int(f.nao) for f in frags), default=0)
    if lockstep is None:          # (an explicit nstreams is a request for that many streams, not for the lock step)
        # (round 4: four fragments of 36 orbitals -- the periodic configs[4] cell -- run 6.8 ms in lock step against 9.5 ms on four streams;
        #  four of ~55, octane BE3, tie at 28-29 ms)
        # (round 5: with the update's parallel regions four fragments of ~55 orbitals -- octane BE3 -- run 26.5 ms in lock step against 28.7 ms on four streams)
        lockstep = nstreams is None and len(frags) >= 4 and nmax <= 64
    if nstreams is None:
        nstreams = 1 if len(frags) <= 1 else (min(6, len(frags)) if nmax <= 96 else min(4 if nmax <= 256 else 2, len(frags)))
        if nstreams > 1 and nmax > 96:
            if mem_free is None:
                mem_free = _device_free_bytes(frags)
            if mem_free is not None:
                work = max(fragment_work_bytes(f.nao, getattr(f, "nsocc", None)) for f in frags)
                nstreams = max(1, min(nstreams, int(0.9 * mem_free // work)))
    return int(nstreams), bool(lockstep)


def _device_free_bytes(frags):
    """free device memory as the fragments' library reports it (qemb_mem_info) plus nothing else: blocks the library has parked count as used,
    so the bound errs on the side of fewer fragments in flight.  None when the fragments are not on a device (host-logic tests)."""
    import ctypes as C
    dev = next((getattr(f, "dev", None) for f in frags if getattr(f, "dev", None) is not None), None)
    lib = getattr(dev, "lib", None)
    if lib is None or not hasattr(lib, "qemb_mem_info"):
        return None
    free_b, total_b = C.c_size_t(), C.c_size_t()
    try:
        if hasattr(lib, "qemb_trim"):
            lib.qemb_trim()                      # parked work space of earlier sweeps is free for this decision
        if lib.qemb_mem_info(C.byref(free_b), C.byref(total_b)) != 0:
            return None
    except Exception:  # noqa: BLE001
        return None
    return float(free_b.value)


def set_cu_partition(lib, parts):
    """Spread the execution contexts that `map_fragments` creates from now on over `parts` disjoint, interleaved sets of compute units (qemb_ctx_partition; 0 or 1:
    every context on the whole chip).  Two parts with four large fragments in flight let the HBM-bound passes of one fragment run beside the MFMA-bound products of
    another: +0.7-2 % on the n = 220 sweep by box (small, launch-bound fragments lose: leave them on the whole chip).  Call it between sweeps: contexts that exist
    are drained and get new streams."""
    from ._lib import check
    check(lib.qemb_ctx_partition(int(parts)), "qemb_ctx_partition", lib)


def map_fragments(fn, frags, nstreams=1):
    """[fn(f) for f in frags], with up to `nstreams` fragments in flight at once: each worker thread is bound to its own
    execution context of the library (HIP stream + workspaces, qemb_ctx_bind), so fragments whose kernels are latency bound
    overlap on the device.  The reference overlaps fragments with a process pool (be_parallel.py:484-513)."""
    frags = list(frags)
    if nstreams <= 1 or len(frags) <= 1:
        return [fn(f) for f in frags]
    from concurrent.futures import ThreadPoolExecutor
    lib = frags[0].dev.lib
    have = lib.qemb_ctx_count(int(nstreams) + 1)          # context 0 stays with the calling thread
    if have < 0:
        from ._lib import check
        check(have, "qemb_ctx_count", lib)
    nwork = min(int(nstreams), have - 1, len(frags))
    if nwork <= 1:
        return [fn(f) for f in frags]
    # STATIC assignment of fragments to worker contexts (longest first, each to the least loaded worker; ties in fragment order): the same
    # fragment meets the same context in every sweep.  The caching allocator of a context hands blocks back by exact size, so a context
    # that gets a fragment of another size than last time misses its pool and goes to the driver (hipMalloc of a solve's ~10^3 buffers:
    # tens of ms) -- with a dynamic queue that happened now and then in any sweep (the 92 ms sweep of the round-3 octane figures,
    # tools/octane_sweep_series.py), with the static assignment only in the first one.
    def cost(f):
        n, o = int(getattr(f, "nao", 0) or 0), getattr(f, "nsocc", None)
        o = n // 2 if o is None else int(o)
        v = max(n - o, 0)
        return float(o * o) * float(v) ** 4 + 4.0 * float(o * v) ** 3
    order = sorted(range(len(frags)), key=lambda k: (-cost(frags[k]), k))
    load, mine = [0.0] * nwork, [[] for _ in range(nwork)]
    for k in order:
        w = min(range(nwork), key=lambda j: (load[j], j))
        mine[w].append(k); load[w] += max(cost(frags[k]), 1.0)

    from ._lib import QEMB_ERR_ALLOC, QembError, check
    res, retry = [None] * len(frags), []

    def work(w):
        check(lib.qemb_ctx_bind(w + 1), "qemb_ctx_bind", lib)
        out = []
        for k in mine[w]:
            try:
                out.append((k, fn(frags[k]), None))
            except QembError as e:
                if getattr(e, "status", None) != QEMB_ERR_ALLOC:
                    raise
                out.append((k, None, e))
        return out
    with ThreadPoolExecutor(max_workers=nwork) as pool:
        for part in pool.map(work, range(nwork)):
            for k, r, err in part:
                if err is not None:
                    retry.append(k)
                else:
                    res[k] = r
    if retry:
        # the working sets of `nwork` fragments did not fit together: what fits one fragment at a time must keep working (the serial sweep
        # is what nstreams = 1 runs); parked blocks of every context are released first
        import warnings
        warnings.warn(f"{len(retry)} fragment(s) ran out of device memory with {nwork} in flight; solving them one at a time", RuntimeWarning, stacklevel=2)
        if hasattr(lib, "qemb_trim"):
            lib.qemb_trim()
        for k in retry:
            res[k] = fn(frags[k])
    return res


def solve_fragments(pot, frags, only_chem=False, opts=None, eeval=False, use_cumulant=True, relax_density=False, nstreams=1, lockstep=False,
                    stats=None):
    """The loop body of be_func (molbe/solver.py:301-547) for the fragments `frags`: update_heff, then the device solve of each.
    nstreams > 1: that many fragments in flight on separate streams (map_fragments).  lockstep: ALL fragments in one library call
    (qemb_frag_solve_batch) -- their CCSD iterations advance together, every operation one grouped launch; the small-fragment regime
    (octane BE2 / BE3), where a fragment alone is bound by its ~110 dependent launches per iteration.  Same results, bit for bit."""
    frags = list(frags)
    if lockstep and len(frags) > 1:
        from .fragsolver import solve_batch
        o_list = []
        for f in frags:
            if pot is not None:
                f.update_heff(pot, only_chem=only_chem)
            assert f.fock is not None and f.heff is not None
            o_list.append(f._solve_inputs(opts, eeval, relax_density))
        # one options struct for the batch: the per-fragment ones differ at most by relax_density, which _solve_inputs set identically
        outs = solve_batch([f.dev for f in frags], [f.nsocc for f in frags], [f.fock + f.heff for f in frags], [f.dm0 for f in frags],
                           opts=o_list[0], eeval=eeval, stats=stats)
        return [f._solve_outputs(out, eeval, use_cumulant) for f, out in zip(frags, outs)]

    def one(fobj):
        if pot is not None:
            fobj.update_heff(pot, only_chem=only_chem)
        assert fobj.fock is not None and fobj.heff is not None
        return fobj.solve(opts=opts, eeval=eeval, use_cumulant=use_cumulant, relax_density=relax_density)
    return map_fragments(one, frags, nstreams)


def be_func(pot, Fobjs, Nocc, solver, enuc, solver_args=None, scratch_dir=None, only_chem=False, eeval=False,
            relax_density=False, return_vec=False, use_cumulant=True, *, opts=None, stats=None, nstreams=1, lockstep=False):
    """molbe/solver.py:244-562 for solver == 'CCSD'.  `opts` (qemb_solver_opts), `stats` (dict collecting per-sweep
    counters), `nstreams` (fragments in flight at once, see map_fragments) and `lockstep` (all fragments in one batched call, see
    solve_fragments) are additions; everything else has the reference's meaning."""
    if solver != "CCSD":
        raise ValueError("Solver not implemented")
    total_e = [0.0, 0.0, 0.0]
    n_iter = 0
    for out in solve_fragments(pot, Fobjs, only_chem, opts, eeval, use_cumulant, relax_density, nstreams, lockstep, stats):
        n_iter += out["n_iter"]
        if eeval:
            total_e = [a + b for a, b in zip(total_e, out["e_frag"])]
    if stats is not None:
        stats["ccsd_iterations"] = stats.get("ccsd_iterations", 0) + n_iter
        stats["fragments"] = stats.get("fragments", 0) + len(Fobjs)
    Ecorr = sum(total_e)
    if eeval and not return_vec:
        return (Ecorr, total_e)
    ernorm, ervec = solve_error(Fobjs, Nocc, only_chem=only_chem)
    if eeval:
        return (ernorm, ervec, [Ecorr, total_e])
    if return_vec:
        return (ernorm, ervec, None)
    return ernorm
