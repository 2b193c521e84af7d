"""Device-memory hygiene: repeated solves (serial, relaxed, multi-stream) must not grow the footprint."""
import sys, ctypes as C, queue
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from helpers import synthetic_fragment
from qemb_oracle import eri
from quemb_amd import _lib
from quemb_amd._lib import check
from quemb_amd.fragsolver import DeviceFragment, default_opts
lib = _lib.init(0)
def free_gb():
    f, t = C.c_size_t(), C.c_size_t(); lib.qemb_mem_info(C.byref(f), C.byref(t)); return f.value / 1e9
frs = []
for i in range(6):
    n, o = 40 + 3 * i, 10 + i
    h, e1 = synthetic_fragment(n, o, 50 + i, scale=0.03)
    fr = DeviceFragment(n, 5, lib=lib); fr.set_eri_s4(eri.pack_s4(e1)); fr.set_energy_data(h, h, None, 1.0, [0, 1])
    frs.append((fr, h, o))
lib.qemb_ctx_count(4)
ids = queue.Queue()
for k in (1, 2, 3): ids.put(k)
pool = ThreadPoolExecutor(max_workers=3, initializer=lambda: check(lib.qemb_ctx_bind(ids.get()), "bind", lib))
def one(t, relax=0):
    fr, h, o = t
    return fr.solve(o, h, opts=default_opts(lib, relax_density=relax), eeval=True)["e_corr_mo"]
hist = []
for rep in range(40):
    e1 = [one(t) for t in frs]
    e2 = list(pool.map(one, frs))
    e3 = [one(t, 1) for t in frs[:2]]
    assert e1 == e2
    hist.append(free_gb())
    print("rep", rep, "free GB %.3f" % hist[-1], flush=True)
# (a block returns to the cache of the context that allocated it, so a fragment visiting a new context may add one block per size
#  class there: bounded by contexts x size classes, not a leak)
print("footprint GB: start %.3f, after 10 reps %.3f, after 40 reps %.3f" % (hist[0], hist[9], hist[-1]))
assert hist[9] - hist[-1] < 2.0, hist
