"""The periodic driver (quemb_amd/kbe_pbe.py, mirror of kbe/pbe.py) against the molecular driver on the SAME system: a ring of nk
cells treated with nk k-points (fragments of the reference cell only) and as one supercell (a fragment per site).  Per unit cell
the HF-in-HF identity, the one-shot CCSD correlation energy and the density-matched one must agree.  Scalar mock here, HIP under
-m gpu (tests/test_gpu_be.py imports `check_periodic_driver`)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hostcheck", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
import kbe_model  # noqa: E402
from kbe_df_source import GammaSourceFromFactor  # noqa: E402


@pytest.fixture(scope="module")
def hlib():
    import build as hc_build
    from quemb_amd import _lib
    return _lib.declare(C.CDLL(str(hc_build.build())))


def drivers(lib, nk=4, nlo=3, seed=5):
    from quemb_amd import eri_transform as et
    from quemb_amd import kbe_pbe, mbe
    from quemb_amd.fragpart import FragPart
    m = kbe_model.build(nk=nk, nlo=nlo, seed=seed)
    ao = et.AOEri(m["eri_super"], m["N"], lib=lib)
    kmf = kbe_pbe.KMeanField(a_vec=m["a_vec"], kpts=m["kpts"], kmesh=m["kmesh"], nelectron=2 * m["nocc_cell"], hcore=m["hk"], S=m["Sk"],
                             mo_coeff=m["Ck"], mo_energy=m["ek"], hf_veff=m["veffk"], e_tot=m["e_tot_cell"])
    kbe = kbe_pbe.BE(kmf, FragPart(**kbe_model.ring_be2_lists(m["N"], nlo)), lib=lib, distribute=False,
                     eri_provider=lambda f: ao.transform(kbe_model.real_space_TA(f, m)))
    mol = mbe.BE(kbe_model.SupercellMF(m), FragPart(**kbe_model.ring_be2_lists(m["N"], m["N"])), lib=lib, distribute=False,
                 schmidt_method="eigh")
    ao.free()
    return m, kbe, mol


def check_periodic_driver(lib, nk=4, nlo=3):
    m, kbe, mol = drivers(lib, nk, nlo)
    # the k-point mean field is the supercell one: the supercell density rebuilt from the k-point orbitals (KFrags.sd) is the oracle's
    from quemb_amd import kbe_pfrag as kp
    ph = kp.get_phase(m["a_vec"], m["kpts"], m["kmesh"])
    Dk = np.stack([c[:, : m["nocc_cell"]] @ c[:, : m["nocc_cell"]].conj().T for c in m["Ck"]])
    Dsup = np.einsum("Rk,kuv,Sk->RuSv", ph, Dk, ph.conj()).reshape(m["N"], m["N"])
    assert np.abs(Dsup - 0.5 * m["mf_super"]["dm"]).max() < 1e-8
    # HF-in-HF: both views reproduce the mean-field energy, per cell the same
    assert abs(kbe.hf_err) < 1e-8 and abs(mol.hf_err) < 1e-8, (kbe.hf_err, mol.hf_err)
    assert abs(kbe.ebe_hf - mol.ebe_hf / nk) < 1e-8
    for f in kbe.Fobjs:
        g = mol.Fobjs[f.AO_in_frag[0]]
        assert f.nao == g.nao and f.nsocc == g.nsocc
        assert abs(f.ebe_hf - g.ebe_hf) < 1e-8
    # one-shot CCSD
    rk, rm = kbe.oneshot(), mol.oneshot()
    assert abs(rk[0] - rm[0] / nk) < 1e-8, (rk[0], rm[0] / nk)
    assert np.abs(np.asarray(rk[1]) - np.asarray(rm[1]) / nk).max() < 1e-8
    # density matching (edge 1-RDM elements + the electron count per cell)
    bk = kbe.optimize(conv_tol=1e-8)
    bm = mol.optimize(conv_tol=1e-8)
    assert abs(kbe.e_corr - mol.e_corr / nk) < 2e-7, (kbe.e_corr, mol.e_corr / nk)
    assert abs(kbe.e_corr - rk[0]) > 1e-6            # the matching moved the energy: the test sees the potentials
    # the matched potentials have the translational symmetry: supercell potential of fragment c = k-point potential of fragment c mod nlo
    pk, pm = np.asarray(kbe.pot), np.asarray(mol.pot)
    assert abs(pk[-1] - pm[-1]) < 1e-5               # chemical potential
    per = (len(pm) - 1) // m["N"]
    for c in range(m["N"]):
        assert np.abs(pm[c * per:(c + 1) * per] - pk[(c % nlo) * per:((c % nlo) + 1) * per]).max() < 1e-5
    return m, kbe, mol, bk, bm


def c5_drivers(lib, rhf=None, with_supercell=True, distribute=False, **model_kw):
    """BASELINE configs[4] at its own dimensions (kbe_model.build_chain: 24 orbitals and 28 electrons per cell, 1 x 1 x 3 k-points, N = 72):
    the periodic driver with the BE2 fragments of the reference cell (4 CH units, each with its two neighbour units: 18 sites, 36 embedding
    orbitals) and fragment ERIs from the density-fitted integrals of the Born-von-Karman supercell through the device CC-GDF transform
    (int_transform='supercell-DF-hip' -> kbe_eri_onthefly.integral_direct_DF), beside the molecular driver on the 72-orbital supercell with
    its 12 fragments and dense integrals."""
    from quemb_amd import kbe_pbe, mbe
    from quemb_amd.fragpart import FragPart
    m = kbe_model.build_chain(rhf=rhf, **model_kw)
    U, u = m["units_per_cell"], m["unit_size"]
    kmf = kbe_pbe.KMeanField(a_vec=m["a_vec"], kpts=m["kpts"], kmesh=m["kmesh"], nelectron=2 * m["nocc_cell"], hcore=m["hk"], S=m["Sk"],
                             mo_coeff=m["Ck"], mo_energy=m["ek"], hf_veff=m["veffk"], e_tot=m["e_tot_cell"])
    kbe = kbe_pbe.BE(kmf, FragPart(**kbe_model.chain_be2_lists(m["n_units"], U, u)), lib=lib, distribute=distribute,
                     int_transform="supercell-DF-hip", df_source=GammaSourceFromFactor(m["B"]))
    mol = None
    if with_supercell:
        mol = mbe.BE(kbe_model.SupercellMF(m), FragPart(**kbe_model.chain_be2_lists(m["n_units"], m["n_units"], u)), lib=lib, distribute=False,
                     schmidt_method="eigh")
    return m, kbe, mol


def check_c5_driver(lib, matching=True, conv_tol=1e-7):
    m, kbe, mol = c5_drivers(lib)
    nk, U = m["nk"], m["units_per_cell"]
    assert m["nlo"] == 24 and nk == 3 and m["nocc_cell"] == 14 and len(kbe.Fobjs) == 4 and len(mol.Fobjs) == 12
    assert all(f.nao == 36 for f in kbe.Fobjs) and all(f.nsocc == g.nsocc for f, g in zip(kbe.Fobjs, mol.Fobjs))
    # HF-in-HF in both views, per cell the same; fragment HF energies of the reference-cell fragments == their supercell twins
    assert abs(kbe.hf_err) < 1e-8 and abs(mol.hf_err) < 1e-8, (kbe.hf_err, mol.hf_err)
    assert abs(kbe.ebe_hf - mol.ebe_hf / nk) < 1e-8
    for c, f in enumerate(kbe.Fobjs):
        assert abs(f.ebe_hf - mol.Fobjs[c].ebe_hf) < 1e-8 and abs(f.ebe_hf - mol.Fobjs[c + U].ebe_hf) < 1e-8
    # the fragment ERIs that came through the CC-GDF transform of the supercell source are the rotated dense integrals
    f0 = kbe.Fobjs[1]
    T = f0.real_space_TA(kbe.a_vec, kbe.kpts, kbe.kmesh)
    assert np.abs(T - kbe_model.real_space_TA(f0, m)).max() < 1e-10
    from qemb_oracle import eri as oeri
    ref = oeri.pack_s4(np.einsum("pqrs,pi,qj,rk,sl->ijkl", m["eri_super"], T, T, T, T, optimize=True))
    assert np.abs(f0.dev.get_eri_s4() - ref).max() < 1e-10
    # one-shot CCSD: E_corr per cell and its three pieces
    rk, rm = kbe.oneshot(), mol.oneshot()
    assert abs(rk[0]) > 1e-3 and abs(rk[0] - rm[0] / nk) < 1e-8, (rk[0], rm[0] / nk)
    assert np.abs(np.asarray(rk[1]) - np.asarray(rm[1]) / nk).max() < 1e-8
    if not matching:
        return m, kbe, mol
    # density matching: 2 x 21 edge elements per fragment + the electron count per cell
    kbe.optimize(conv_tol=conv_tol)
    mol.optimize(conv_tol=conv_tol)
    assert abs(kbe.e_corr - mol.e_corr / nk) < 5e-7, (kbe.e_corr, mol.e_corr / nk)
    assert abs(kbe.e_corr - rk[0]) > 1e-6
    pk, pm = np.asarray(kbe.pot), np.asarray(mol.pot)
    assert abs(pk[-1] - pm[-1]) < 2e-5
    per = (len(pm) - 1) // m["n_units"]
    assert per == 42 and len(pk) == 4 * per + 1
    for c in range(m["n_units"]):
        assert np.abs(pm[c * per:(c + 1) * per] - pk[(c % U) * per:((c % U) + 1) * per]).max() < 5e-5
    return m, kbe, mol


def check_gamma_point_driver_with_direct_df(lib):
    """nk = 1 (the supercell as the unit cell): int_transform='int-direct-DF-hip' through kbe_eri_onthefly on a source that reproduces
    the model's DF factor == the molecular driver on the dense integrals."""
    from quemb_amd import kbe_pbe, mbe
    from quemb_amd.fragpart import FragPart
    m = kbe_model.build(nk=3, nlo=2, seed=2)
    N = m["N"]
    mf = m["mf_super"]
    kmf = kbe_pbe.KMeanField(a_vec=np.diag([3 * m["a"], 12.0, 12.0]), kpts=np.zeros((1, 3)), kmesh=[1, 1, 1], nelectron=2 * m["nocc"],
                             hcore=m["h_super"][None].astype(np.complex128), S=np.eye(N, dtype=np.complex128)[None],
                             mo_coeff=mf["mo_coeff"][None].astype(np.complex128), mo_energy=mf["mo_energy"][None],
                             hf_veff=m["veff_super"][None].astype(np.complex128), e_tot=mf["e_tot"])
    fp = FragPart(**kbe_model.ring_be2_lists(N, N))
    kbe = kbe_pbe.BE(kmf, fp, lib=lib, distribute=False, int_transform="int-direct-DF-hip", df_source=GammaSourceFromFactor(m["B"]))
    mol = mbe.BE(kbe_model.SupercellMF(m), FragPart(**kbe_model.ring_be2_lists(N, N)), lib=lib, distribute=False, schmidt_method="eigh")
    assert abs(kbe.hf_err) < 1e-8 and abs(kbe.ebe_hf - mol.ebe_hf) < 1e-8
    rk, rm = kbe.oneshot(), mol.oneshot()
    assert abs(rk[0] - rm[0]) < 1e-8 and abs(rk[0]) > 1e-3


def test_periodic_driver_matches_the_supercell_on_the_mock(hlib):
    check_periodic_driver(hlib)


def test_periodic_driver_argument_errors(hlib):
    from quemb_amd import kbe_pbe
    from quemb_amd.fragpart import FragPart
    m = kbe_model.build(nk=3, nlo=2, seed=8)
    kmf = kbe_pbe.KMeanField(a_vec=m["a_vec"], kpts=m["kpts"], kmesh=m["kmesh"], nelectron=2, hcore=m["hk"], S=m["Sk"], mo_coeff=m["Ck"],
                             mo_energy=m["ek"], hf_veff=m["veffk"], e_tot=m["e_tot_cell"])
    fp = FragPart(**kbe_model.ring_be2_lists(m["N"], 2))
    with pytest.raises(NotImplementedError, match="k-point sampled ERI not implemented for int-direct-DF"):
        kbe_pbe.BE(kmf, fp, lib=hlib, distribute=False, int_transform="int-direct-DF-hip")
    with pytest.raises(ValueError, match="eri_provider"):
        kbe_pbe.BE(kmf, fp, lib=hlib, distribute=False)
    with pytest.raises(ValueError, match="int_transform"):
        kbe_pbe.BE(kmf, fp, lib=hlib, distribute=False, int_transform="out-core-DF")


def test_gamma_point_driver_with_direct_df_on_the_mock(hlib):
    check_gamma_point_driver_with_direct_df(hlib)


@pytest.mark.timeout(900)
def test_c5_dimensions_kpoint_view_equals_supercell_view_on_the_mock(hlib):
    """BASELINE configs[4] at its own size (24 orbitals per cell x 3 k-points), one-shot part; the density matching at this size runs on
    the GPU (tests/test_gpu_be.py) -- minutes on the scalar mock"""
    check_c5_driver(hlib, matching=False)


# ---- world_size 2 over gloo: the k-point fragments of the reference cell sharded over two ranks, one all-reduce per sweep
def _worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import build as hc_build
    from quemb_amd import _lib
    from quemb_amd import eri_transform as et
    from quemb_amd import kbe_pbe
    from quemb_amd.fragpart import FragPart
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    m = kbe_model.build(nk=4, nlo=3, seed=5)
    ao = et.AOEri(m["eri_super"], m["N"], lib=lib)
    kmf = kbe_pbe.KMeanField(a_vec=m["a_vec"], kpts=m["kpts"], kmesh=m["kmesh"], nelectron=2 * m["nocc_cell"], hcore=m["hk"], S=m["Sk"],
                             mo_coeff=m["Ck"], mo_energy=m["ek"], hf_veff=m["veffk"], e_tot=m["e_tot_cell"])
    be = kbe_pbe.BE(kmf, FragPart(**kbe_model.ring_be2_lists(m["N"], 3)), lib=lib, distribute=True,
                    eri_provider=lambda f: ao.transform(kbe_model.real_space_TA(f, m)))
    assert be.world == world and sorted(set(be.owner)) == list(range(world))
    assert all((be.Fobjs[i].fock is not None) == (be.owner[i] == rank) for i in range(3))
    e1 = be.oneshot()[0]
    be.optimize(conv_tol=1e-8)
    q.put((rank, be.hf_err, e1, be.e_corr, [float(x) for x in be.pot]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_periodic_driver_two_ranks_equal_one(hlib):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    _, kbe, _ = drivers(hlib)
    e1 = kbe.oneshot()[0]
    kbe.optimize(conv_tol=1e-8)
    for (_, hf_err, e_one, e_opt, pot) in res:
        assert abs(hf_err) < 1e-8 and abs(e_one - e1) < 1e-10 and abs(e_opt - kbe.e_corr) < 1e-9
        assert np.allclose(pot, kbe.pot, atol=1e-8)
    assert res[0][4] == res[1][4]                 # bit-identical potentials on both ranks


def _c5_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import build as hc_build
    from quemb_amd import _lib
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    m, be, _ = c5_drivers(lib, with_supercell=False, distribute=True)
    assert be.world == world and [be.owner.count(r) for r in range(world)] == [2, 2]
    assert all((be.Fobjs[i].fock is not None) == (be.owner[i] == rank) for i in range(4))
    e1 = be.oneshot()[0]
    be.optimize(conv_tol=1e-7)
    q.put((rank, be.hf_err, e1, be.e_corr, [float(x) for x in be.pot]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_c5_dimensions_two_ranks_equal_one(hlib):
    """configs[4]'s sharding at its own size on the mock: the four reference-cell fragments (36 embedding orbitals each) on two gloo ranks,
    one all-reduce of the 169-slot residual buffer per sweep; HF-in-HF, one-shot and density-matched energies and potentials of one rank"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c5_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    _, kbe, _ = c5_drivers(hlib, with_supercell=False)          # the one-rank run, meanwhile
    e1 = kbe.oneshot()[0]
    kbe.optimize(conv_tol=1e-7)
    res = sorted((q.get(timeout=800) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for (_, hf_err, e_one, e_opt, pot) in res:
        assert abs(hf_err) < 1e-8 and abs(e_one - e1) < 1e-10 and abs(e_opt - kbe.e_corr) < 1e-9
        assert np.allclose(pot, kbe.pot, atol=1e-8)
    assert res[0][4] == res[1][4]                 # bit-identical potentials on both ranks
