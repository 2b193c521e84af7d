"""be_func_parallel -- the fragment sweep sharded over GPUs (one process per GPU).

Reference: molbe/be_parallel.py:413-553 farms `run_solver` out to a `pathos.ProcessPool(nproc // ompnum)`
(:484-513) and pickles (e_f, mo_coeff, rdm1, rdm2s, rdm1_tmp) back through pipes (:517) -- rdm2s is n^4 doubles.
Here fragments are statically partitioned over ranks (longest-processing-time by the o^2 v^4 ladder cost, the
role of `order_by_size`, molbe/fragment.py:68-70), every rank keeps its fragments' ERIs resident on its GPU, and
the ONLY exchange per sweep is one sum-all-reduce (RCCL over xGMI) of the residual buffer
    [edge_vals (n_match), cen_vals (n_match), sum centre diag, e1, e2, ec, n_iter]
where each rank writes the slots its fragments own (ErrorMap) and zeros elsewhere -- a few kB, latency bound.

Transport, in order of preference:
  1. the library's own communicator (`quemb_amd.comm`, C ABI qemb_comm_*: ncclAllReduce on a persistent RCCL communicator and the
     library's stream) -- no Python dependency beyond ctypes; `comm.init_from_env(lib)` after `_lib.init(LOCAL_RANK)`;
  2. an initialised `torch.distributed` process group (gloo on CPU -- tests/test_distributed_gloo.py -- or nccl), for callers that
     already run under torch.  With torch AND libqemb_hip.so in one process, import torch first: both bring a HIP runtime under one
     SONAME; loaded in that order they share torch's, the other way round `torch.cuda` finds no device afterwards (measured on this pool).
"""

from __future__ import annotations

import numpy as np

from .solver import ErrorMap


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _libcomm():
    from . import comm
    return comm.active()


def world():
    c = _libcomm()
    if c:
        return c[1], c[2]
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def partition_fragments(costs, world_size):
    """Static LPT assignment: heaviest fragment first onto the currently lightest rank.  Returns owner[frag]."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def fragment_cost(n, o):
    v = n - o
    return float(o * o) * float(v) ** 4 + 4.0 * float(o * v) ** 3


class RankFailure(RuntimeError):
    """Raised on EVERY rank when at least one rank failed while producing its share of an all-reduced buffer."""


def all_reduce_sum(buf: np.ndarray, device=None, error: BaseException | None = None, force: bool = False):
    """In-place sum over ranks of a float64 numpy buffer (no-op for a single process).

    `error` is the exception this rank caught while filling `buf` (None if it succeeded).  A fragment failure (SCF / CCSD
    non-convergence, allocation failure, ...) is local to one rank; if that rank simply raised, the others would wait in the
    collective forever.  So the failure count travels in one extra slot of the SAME all-reduce and every rank raises after it."""
    c = _libcomm()
    d = None if c else _dist()
    if c is None and (d is None or (d.get_world_size() == 1 and not force)):      # force: run the collective also on a one-rank group (tests)
        if error is not None:
            raise error
        return buf
    ext = np.append(np.asarray(buf, dtype=np.float64).ravel(), 0.0 if error is None else 1.0)
    if error is not None:
        ext[:-1] = 0.0
    if c is not None:                       # the library's RCCL communicator (qemb_comm_allreduce)
        from . import comm
        comm.all_reduce(c[0], ext, comm.SUM)
        rank_, ws_ = c[1], c[2]
    else:                                   # a torch.distributed group the caller created
        import torch
        rank_, ws_ = d.get_rank(), d.get_world_size()
        if d.get_backend() == "nccl":
            if device is None:
                from . import _lib
                idx = _lib._initialised_device if _lib._initialised_device is not None else torch.cuda.current_device()
                device = torch.device("cuda", idx)
            t = torch.from_numpy(ext).to(device)
            d.all_reduce(t, op=d.ReduceOp.SUM)
            ext = t.cpu().numpy()
        else:
            t = torch.from_numpy(ext)
            d.all_reduce(t, op=d.ReduceOp.SUM)
    nfail = int(round(ext[-1]))
    if nfail:
        msg = f"{nfail} of {ws_} rank(s) failed in this step"
        if error is not None:
            raise RankFailure(f"{msg}; rank {rank_}: {error}") from error
        raise RankFailure(msg + " (this rank succeeded)")
    buf.reshape(-1)[:] = ext[:-1]
    return buf


def all_reduce_bytes(n_values: int) -> int:
    """bytes one rank contributes to the all-reduce of an n-value residual buffer (the values + the failure slot)"""
    return 8 * (int(n_values) + 1)


def be_func_parallel(pot, Fobjs, Nocc, solver, enuc, solver_args=None, scratch_dir=None, only_chem=False, eeval=False,
                     relax_density=False, return_vec=False, use_cumulant=True, nproc=1, ompnum=1, *, owner=None, opts=None,
                     stats=None, emap=None, nstreams=1, lockstep=False):
    """Same return contract as be_func (molbe/be_parallel.py:413-553).  `Fobjs` is the full fragment list on every
    rank; only the fragments with owner[i] == rank need device state (fock / ERIs) on this rank."""
    if solver != "CCSD":
        raise ValueError("Solver not implemented")
    rank, ws = world()
    if owner is None:
        owner = [i % ws for i in range(len(Fobjs))]
    mine = [i for i in range(len(Fobjs)) if owner[i] == rank]
    emap = emap or ErrorMap(Fobjs)
    nm = emap.n_match
    buf = np.zeros(2 * nm + 5)
    from .solver import solve_fragments
    err = None
    try:
        for out in solve_fragments(pot, [Fobjs[i] for i in mine], only_chem, opts, eeval, use_cumulant, relax_density, nstreams, lockstep, stats):
            buf[2 * nm + 4] += out["n_iter"]
            if eeval:
                buf[2 * nm + 1: 2 * nm + 4] += out["e_frag"]
        buf[2 * nm] = emap.fill(Fobjs, mine, buf[:nm], buf[nm:2 * nm])
    except Exception as e:  # noqa: BLE001 -- carried through the collective, re-raised on every rank
        err = e
    all_reduce_sum(buf, error=err)
    if stats is not None:
        stats["ccsd_iterations"] = stats.get("ccsd_iterations", 0) + int(round(buf[2 * nm + 4]))
        stats["fragments"] = stats.get("fragments", 0) + len(Fobjs)
        stats["fragments_this_rank"] = len(mine)
        stats["allreduce_bytes_per_sweep"] = all_reduce_bytes(len(buf))
    total_e = [float(x) for x in buf[2 * nm + 1: 2 * nm + 4]]
    Ecorr = sum(total_e)
    if eeval and not return_vec:
        return (Ecorr, total_e)
    tr = buf[2 * nm] / Fobjs[0].unitcell_nkpt
    if only_chem:
        err = tr - Nocc
        ernorm, ervec = abs(err), np.asarray([err])
    else:
        ervec = np.append(buf[:nm], tr) - np.append(buf[nm:2 * nm], Nocc)
        ernorm = float(np.mean(ervec * ervec) ** 0.5)
    if eeval:
        return (ernorm, ervec, [Ecorr, total_e])
    if return_vec:
        return (ernorm, ervec, None)
    return ernorm
