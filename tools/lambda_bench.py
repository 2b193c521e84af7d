"""Time the relaxed-density path (Lambda equations + response densities) on one synthetic fragment."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from quemb_amd import _lib
from quemb_amd.fragsolver import DeviceFragment, default_opts
from frag_bench import synthetic_on_device

lib = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
o = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nf = max(1, n // 10)
h, d4, _dB, _naux = synthetic_on_device(lib, n, 20260803); _dB.free()
fr = DeviceFragment(n, nf, lib=lib)
fr.set_eri_s4_dev(d4.ptr)
rng = np.random.default_rng(1)
h1 = rng.standard_normal((n, n)); h1 = h1 + h1.T
fr.set_energy_data(h1, h1, None, 1.0, list(range(nf)))
for relax in (0, 1, 1):
    t = time.time()
    out = fr.solve(o, h, opts=default_opts(lib, relax_density=relax, verbose=int(len(sys.argv) > 3)), eeval=True)
    lib.qemb_sync()
    print(json.dumps(dict(n=n, o=o, relax_density=relax, wall_s=time.time() - t, ccsd_iters=out["n_iter"], lambda_iters=out["lambda_iters"],
                          e_frag=[float(x) for x in out["e_frag"]], trace_rdm1=float(np.trace(out["rdm1_mo"])))), flush=True)
