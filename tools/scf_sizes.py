"""GPU: wall time of the fragment RHF alone (cold from the core guess, then warm from the previous density with a 1e-3 change of h)
at n = 220, 260, 300 -- the Jacobi rounds are one launch each above n = 96, so this is where their launch count shows."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from quemb_amd import _lib
from tools.frag_bench import synthetic_on_device
from quemb_amd.fragsolver import DeviceFragment, default_opts
lib = _lib.init(0)
for n, o in ((220, 20), (260, 26), (300, 30)):
    h, d4, _dB, _naux = synthetic_on_device(lib, n, 20260803); _dB.free()
    fr = DeviceFragment(n, 22); fr.set_eri_s4_dev(d4.ptr); d4.free()
    t = time.time(); r = fr.scf(o, h, None, opts=default_opts(verbose=1)); lib.qemb_sync(); t1 = time.time() - t
    C = r["mo_coeff"]; dm0 = 2 * C[:, :o] @ C[:, :o].T
    h2 = h.copy(); h2[:4, :4] += 1e-3
    t = time.time(); r2 = fr.scf(o, h2, dm0, opts=default_opts(verbose=1)); lib.qemb_sync(); t2 = time.time() - t
    print("n", n, "cold scf s", round(t1, 3), "cycles", r["cycles"], "warm s", round(t2, 3), "cycles", r2["cycles"], flush=True)
    fr.free(); lib.qemb_trim()
