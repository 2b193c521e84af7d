"""Sweep-by-sweep times of the octane BE2 one-shot sweep from a cold start, with Python's garbage collector observed (round 4: where
did the 92 ms sweep of the round-3 bench line come from?).  Prints, per mode, the time of every sweep from the very first one and every
collection of the cyclic GC that fell inside a timed sweep (generation, duration).

    python tools/octane_sweep_series.py [sweeps]        (results on stderr as RESULT lines)
"""
import gc
import sys
import time

sys.path.insert(0, ".")
from pathlib import Path

from quemb_amd import _lib
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE

lib = _lib.init(0)
G = Path("tests/golden")
nsweep = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
# ballast: the host objects a bench process holds by the time it reaches this section (the n = 220 ring: arrays, ctypes objects, dicts)
ballast = [dict(a=[float(i), str(i)], b=(i, [i])) for i in range(400000)] if "--ballast" in sys.argv else None

pauses = []
t_gc = [0.0]


def on_gc(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        pauses.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(on_gc)
for kw in (dict(nstreams=6, lockstep=False), dict(lockstep=True), dict(nstreams=6, lockstep=False), dict(lockstep=True)):      # each mode twice: the second BE object of a mode finds the pools the other mode left
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False, lib=lib, **kw)
    ts, hits, misses = [], [], []
    import ctypes as C
    for k in range(nsweep):
        n0 = len(pauses)
        lib.qemb_alloc_stats(None, None, None, None, 1)
        lib.qemb_device_sync(); t0 = time.perf_counter()
        be.oneshot()
        lib.qemb_device_sync(); ts.append((time.perf_counter() - t0) * 1e3)
        nm, nf, ms_, gb = C.c_longlong(), C.c_longlong(), C.c_double(), C.c_double()
        lib.qemb_alloc_stats(C.byref(nm), C.byref(nf), C.byref(ms_), C.byref(gb), 0)
        misses.append("%d/%.1fms" % (nm.value, ms_.value))
        for g, ms in pauses[n0:]:
            hits.append((k, g, round(ms, 2)))
    print("RESULT", kw, "sweeps from cold:", " ".join("%.1f" % t for t in ts), file=sys.stderr, flush=True)
    print("RESULT", kw, "driver allocations (pool misses) per sweep, count/host ms:", " ".join(misses), file=sys.stderr, flush=True)
    print("RESULT", kw, "GC collections inside timed sweeps (sweep, generation, ms):", hits, file=sys.stderr, flush=True)
t0 = time.perf_counter(); gc.collect(); print("RESULT one full collection now: %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
