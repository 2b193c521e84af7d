"""Wall time of the octane/STO-3G BE2 pipeline on the GPU (BASELINE configs[1]): init, one-shot sweep, density matching."""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
t = time.time(); mol = Mole(G / "octane.xyz"); mf = RHF(mol); mf.kernel(); print("host integrals + RHF s", time.time() - t, flush=True)
t = time.time(); be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False); print("BE init (Schmidt, ERI transform, fragment HF) s", time.time() - t, flush=True)
for k in range(3):
    t = time.time(); e, _ = be.oneshot(); print("one-shot sweep s", time.time() - t, "iters", be.stats, flush=True)
t = time.time(); opt = be.optimize(solver="CCSD"); print("density matching s", time.time() - t, "objfunc calls", opt.n_objfunc, "QN iters", opt.iter, "E_corr", be.e_corr, flush=True)
