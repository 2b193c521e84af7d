"""CPU, world_size 2 over gloo: the sharded fragment sweep (be_func_parallel) gives the same energies, residual vector
and optimised potentials as the single-process sweep.  The device layer is the scalar mock (tests/hostcheck)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import build as hc_build
    from quemb_amd import _lib
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    mol = Mole([["H", (0.0, 0.0, float(i))] for i in range(8)])
    mf = RHF(mol); mf.kernel()
    fobj = FragPart.from_json(ROOT / "tests" / "golden" / "fragmentation.json", "test_autogen_h_linear_be2")
    be = BE(mf, fobj, lib=lib, distribute=True)
    assert be.world == world and sorted(set(be.owner)) == list(range(world))
    assert all((be.Fobjs[i].fock is not None) == (be.owner[i] == rank) for i in range(fobj.n_frag))
    ecorr, comps = be.oneshot()
    opt = be.optimize(solver="CCSD", only_chem=False, conv_tol=1e-7)
    D_ao = be.rdm1_fullbasis(only_rdm1=True)                              # summed over ranks
    Jn = be.compute_numerical_jacobian("CCSD", False, 1, step_size=1e-4)   # every rank fills the columns of its fragments
    q.put((rank, ecorr, list(comps), be.ebe_hf, list(be.pot), be.e_corr, opt.err, opt.iter, D_ao, Jn))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sweep_equals_single_process():
    import ctypes as C
    import torch.multiprocessing as mp
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck"):
        sys.path.insert(0, str(p))
    import build as hc_build
    hc_build.build()                      # build the mock device library once, before the ranks start
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process reference
    import build as hc_build
    from quemb_amd import _lib
    from quemb_amd.fragpart import FragPart
    from quemb_amd.integrals import RHF, Mole
    from quemb_amd.mbe import BE
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    mol = Mole([["H", (0.0, 0.0, float(i))] for i in range(8)])
    mf = RHF(mol); mf.kernel()
    fobj = FragPart.from_json(ROOT / "tests" / "golden" / "fragmentation.json", "test_autogen_h_linear_be2")
    be = BE(mf, fobj, lib=lib, distribute=False)
    e1, c1 = be.oneshot()
    opt = be.optimize(solver="CCSD", only_chem=False, conv_tol=1e-7)
    D1 = be.rdm1_fullbasis(only_rdm1=True)
    J1 = be.compute_numerical_jacobian("CCSD", False, 1, step_size=1e-4)
    for (rank, ecorr, comps, ebe_hf, pot, e_opt, err, it, D_ao, Jn) in res:
        assert np.abs(D_ao - D1).max() < 1e-8 and np.abs(Jn - J1).max() < 1e-6
        assert abs(ecorr - e1) < 1e-11 and np.allclose(comps, c1, atol=1e-11)
        assert abs(ebe_hf - be.ebe_hf) < 1e-10
        assert np.allclose(pot, be.pot, atol=1e-8) and abs(e_opt - be.e_corr) < 1e-9
        assert err < 1e-7 and it == opt.iter
    # both ranks hold bit-identical potentials (same all-reduced residual -> same host QN on every rank)
    assert res[0][4] == res[1][4]


def test_lpt_partition():
    from quemb_amd.be_parallel import fragment_cost, partition_fragments
    costs = [fragment_cost(42, 21)] * 4 + [fragment_cost(40, 22)] * 2
    own = partition_fragments(costs, 2)
    assert sorted(own.count(r) for r in range(2)) == [3, 3]
    own8 = partition_fragments([1.0] * 64, 8)
    assert all(own8.count(r) == 8 for r in range(8))


# ---- world size 4, fragments of unequal cost: LPT balance, identical results, and a failing rank that must not hang the others ----
HET_SIZES = [(16, 5), (9, 3), (9, 3), (8, 3), (8, 3), (8, 3), (7, 2), (7, 2), (6, 2), (6, 2), (6, 2)]   # (n, n_occ): one large + many small


def _het_fragments(lib):
    """A ring of synthetic fragments of unequal size: fragment I's edge AOs [0, 1] are matched against the centre AOs [2, 3] of
    fragment I+1 (the index structure solve_error reads, molbe/solver.py:724-766); every fragment carries its own ERIs / Fock."""
    from helpers import synthetic_fragment
    from qemb_oracle import eri as oeri
    from quemb_amd.pfrag import Frags
    F = len(HET_SIZES)
    frs = []
    for I, (n, o) in enumerate(HET_SIZES):
        f = Frags(list(range(4)), I, [[0, 1]], [(I + 1) % F], [[0, 1]], [[2, 3]], (1.0, [2, 3]), [2, 3], lib=lib)
        h, e1 = synthetic_fragment(n, o, 4000 + I)
        rng = np.random.default_rng(I)
        mk = lambda: (lambda a: 0.05 * (a + a.T))(rng.standard_normal((n, n)))
        f.set_eri(oeri.pack_s4(e1))
        f.nao, f.nsocc, f.h1, f.veff0, f.veff, f.fock, f.heff, f.dm0 = n, o, mk(), mk(), mk(), h, np.zeros((n, n)), None
        frs.append(f)
    c = 0
    for f in frs:
        f.udim = c
        c = f.set_udim(c)
    return frs, c + 1


def _het_worker(rank, world, port, q, fail_rank):
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    import build as hc_build
    from quemb_amd import _lib
    from quemb_amd.be_parallel import RankFailure, be_func_parallel, fragment_cost, partition_fragments
    from quemb_amd.fragsolver import default_opts
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    costs = [fragment_cost(n, o) for n, o in HET_SIZES]
    owner = partition_fragments(costs, world)
    # every rank builds the full list (host data only matters for its own fragments; ERIs of the others are never touched)
    frs, npot = _het_fragments(lib)
    pot = list(0.01 * np.sin(np.arange(npot)))
    stats = {}
    err, vec, (ecorr, comps) = be_func_parallel(pot, frs, 7, "CCSD", 0.0, eeval=True, return_vec=True, owner=owner, stats=stats)
    out = dict(rank=rank, err=err, vec=vec, ecorr=ecorr, comps=list(comps), owner=owner, stats=dict(stats))
    # a fragment failure on ONE rank (CCSD cannot converge in one cycle): every rank must get RankFailure out of the same collective
    opts = default_opts(lib)
    opts_fail = default_opts(lib, cc_max_cycle=1)
    per_frag = [opts_fail if owner[i] == fail_rank else opts for i in range(len(frs))]
    try:
        orig = [f.solve for f in frs]              # be_func_parallel passes ONE opts object: pick per fragment through solve()
        for f, o_ in zip(frs, per_frag):
            f.solve = (lambda s, oo: (lambda opts=None, **kw: s(opts=oo, **kw)))(f.solve, o_)
        be_func_parallel(pot, frs, 7, "CCSD", 0.0, eeval=True, return_vec=True, owner=owner)
        out["failure"] = "no exception"
    except RankFailure as e:
        out["failure"] = "RankFailure"
        out["failure_msg"] = str(e)
    finally:
        for f, s in zip(frs, orig):
            f.solve = s
    # and the communicator is still usable afterwards
    err2, vec2, _ = be_func_parallel(pot, frs, 7, "CCSD", 0.0, eeval=True, return_vec=True, owner=owner)
    out["err_after"] = err2
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_four_ranks_heterogeneous_fragments_balance_and_failure_propagation():
    import ctypes as C
    import torch.multiprocessing as mp
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    import build as hc_build
    hc_build.build()
    from quemb_amd.be_parallel import fragment_cost, partition_fragments
    world = 4
    costs = [fragment_cost(n, o) for n, o in HET_SIZES]
    owner = partition_fragments(costs, world)
    load = [sum(c for c, r in zip(costs, owner) if r == k) for k in range(world)]
    # LPT: the heaviest fragment sits alone on its rank and nothing else can exceed it; the other ranks share the rest evenly
    big = max(costs)
    assert owner.count(owner[costs.index(big)]) == 1 and max(load) == big
    rest = sorted(load)[:-1]
    assert max(rest) <= 1.35 * (sum(rest) / len(rest)), load
    assert all(owner.count(k) >= 1 for k in range(world))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    fail_rank = owner[3]
    procs = [ctx.Process(target=_het_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=800) for _ in range(world)), key=lambda r: r["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference through be_func
    from quemb_amd import _lib
    from quemb_amd.solver import be_func
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    frs, npot = _het_fragments(lib)
    pot = list(0.01 * np.sin(np.arange(npot)))
    err1, vec1, (e1, c1) = be_func(pot, frs, 7, "CCSD", 0.0, eeval=True, return_vec=True)
    for r in res:
        assert r["owner"] == owner
        assert abs(r["err"] - err1) < 1e-12 and np.abs(r["vec"] - vec1).max() < 1e-12
        assert abs(r["ecorr"] - e1) < 1e-11 and np.allclose(r["comps"], c1, atol=1e-11)
        assert r["stats"]["fragments_this_rank"] == owner.count(r["rank"])
        assert r["stats"]["allreduce_bytes_per_sweep"] == 8 * (len(vec1) - 1) * 2 + 8 * 5 + 8
        assert r["failure"] == "RankFailure", r
        assert ("this rank succeeded" in r["failure_msg"]) == (r["rank"] != fail_rank)
        assert abs(r["err_after"] - err1) < 1e-12
    assert all(np.array_equal(res[0]["vec"], r["vec"]) for r in res[1:])      # bit-identical residual on every rank
