"""Repro of the slow lock-step sweep (round 4): two BE objects in lock-step mode one after the other, a few sweeps each, every sweep timed;
QEMB_BATCH_TRACE=1 prints the phases.  Under rocprofv3 --kernel-trace (QEMB_GRAPH=1) tools/trace_gaps.py then shows what the device did."""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd import _lib
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
lib = _lib.init(0)
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
modes = sys.argv[1:] or ["lockstep", "lockstep", "streams", "lockstep"]
for m in modes:
    kw = dict(lockstep=True) if m == "lockstep" else dict(nstreams=6, lockstep=False)
    import ctypes as C
    def stats(tag):
        nm, nf, ms_, gb = C.c_longlong(), C.c_longlong(), C.c_double(), C.c_double()
        lib.qemb_alloc_stats(C.byref(nm), C.byref(nf), C.byref(ms_), C.byref(gb), 1)
        print("ALLOC %-22s driver mallocs %4d  frees %4d  host ms %.1f  GB %.3f" % (tag, nm.value, nf.value, ms_.value, gb.value), file=sys.stderr, flush=True)
    stats("before BE")
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False, lib=lib, **kw)
    stats("BE constructed")
    ts = []
    for k in range(5):
        lib.qemb_device_sync(); t0 = time.perf_counter()
        be.oneshot()
        lib.qemb_device_sync(); ts.append((time.perf_counter() - t0) * 1e3)
        stats("%s sweep %d (%.1f ms)" % (m, k, ts[-1]))
    print("RESULT", m, " ".join("%.1f" % t for t in ts), file=sys.stderr, flush=True)
