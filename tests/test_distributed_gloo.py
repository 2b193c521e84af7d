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
