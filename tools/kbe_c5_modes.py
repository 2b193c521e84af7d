"""configs[4]-sized periodic sweep (bench.kbe_c5_sweeps' system) in the three sweep modes: serial, four streams, lock step."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bench
from quemb_amd import _lib
lib = _lib.init(0)
import kbe_model
from kbe_df_source import GammaSourceFromFactor
from quemb_amd import kbe_pbe
from quemb_amd.fragpart import FragPart
from quemb_amd.fragsolver import DeviceFragment


def device_rhf(hs, eri_s1, nocc):
    N = hs.shape[0]; il = np.tril_indices(N)
    fr = DeviceFragment(N, N, lib=lib); fr.set_eri_s4(np.ascontiguousarray(eri_s1[il[0], il[1]][:, il[0], il[1]]))
    r = fr.scf(nocc, hs, None); C_ = r["mo_coeff"]; dm = 2.0 * C_[:, :nocc] @ C_[:, :nocc].T
    J, K = fr.jk(dm); fr.free(); veff = J - 0.5 * K
    return dict(mo_coeff=C_, mo_energy=r["mo_energy"], dm=dm, e_tot=0.5 * float(np.einsum("ij,ji->", 2.0 * hs + veff, dm)), veff=veff)


m = kbe_model.build_chain(rhf=device_rhf)
kmf = kbe_pbe.KMeanField(a_vec=m["a_vec"], kpts=m["kpts"], kmesh=m["kmesh"], nelectron=2 * m["nocc_cell"], hcore=m["hk"], S=m["Sk"],
                         mo_coeff=m["Ck"], mo_energy=m["ek"], hf_veff=m["veffk"], e_tot=m["e_tot_cell"])
for label, kw in (("serial", dict(nstreams=1, lockstep=False)), ("streams4", dict(nstreams=4, lockstep=False)), ("lockstep", dict(lockstep=True))):
    be = kbe_pbe.BE(kmf, FragPart(**kbe_model.chain_be2_lists(m["n_units"], m["units_per_cell"], m["unit_size"])), lib=lib, distribute=False,
                    int_transform="supercell-DF-hip", df_source=GammaSourceFromFactor(m["B"]), **kw)
    for _ in range(3): be.oneshot()
    ts, (e, _) = bench.timed_sweeps(lib, be.oneshot, 16)
    st = bench._stats_ms(ts)
    print("RESULT %-9s p50 %.2f  p95 %.2f  max %.2f ms   E_corr/cell %.12f" % (label, st["p50_ms"], st["p95_ms"], st["max_ms"], e), file=sys.stderr, flush=True)
