"""Does a second / third fragment in flight help at n = 220?  8 synthetic fragments, full solves, 1..3 streams."""
import sys, time, queue
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from quemb_amd import _lib
from quemb_amd._lib import check
from quemb_amd.fragsolver import DeviceFragment, default_opts
from frag_bench import synthetic_on_device
lib = _lib.init(0)
n, o, F = 220, 20, 8
frs = []
for i in range(F):
    h, d4, _dB, _naux = synthetic_on_device(lib, n, 20260803 + i); _dB.free()
    fr = DeviceFragment(n, 22, lib=lib); fr.set_eri_s4_dev(d4.ptr); d4.free()
    r = fr.scf(o, h, None)
    frs.append((fr, h, 2.0 * r["mo_coeff"][:, :o] @ r["mo_coeff"][:, :o].T))
opts = default_opts(lib)
def one(t):
    fr, h, dm0 = t
    return fr.solve(o, h, dm0, opts=opts, eeval=False)["n_iter"]
for ns in (1, 2, 3):
    if ns > 1:
        have = lib.qemb_ctx_count(ns + 1)
        ids = queue.Queue()
        for k in range(1, ns + 1): ids.put(k)
        pool = ThreadPoolExecutor(max_workers=ns, initializer=lambda: check(lib.qemb_ctx_bind(ids.get()), "bind", lib))
        run = lambda: sum(pool.map(one, frs))
    else:
        run = lambda: sum(one(t) for t in frs)
    run()
    t = time.time(); nit = run(); lib.qemb_sync(); dt = time.time() - t
    print("RESULT nstreams=%d sweep %.3f s  %d iterations  %.2f it/s" % (ns, dt, nit, nit / dt), flush=True)
