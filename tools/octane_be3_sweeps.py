"""octane/STO-3G BE3 one-shot sweep (the review's second small-fragment target: <= 15 ms): serial, six streams, lock step -- median of five sweeps each."""
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
for label, kw in (("serial", dict(nstreams=1)), ("streams6", dict(nstreams=6)), ("lockstep", dict(lockstep=True))):
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be3"), distribute=False, lib=lib, **kw)
    be.oneshot()
    ts = []
    for _ in range(5):
        lib.qemb_device_sync(); t0 = time.perf_counter()
        e, _ = be.oneshot()
        lib.qemb_device_sync(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("RESULT octane_be3 %s: median %.1f ms  min %.1f ms  E_corr %.12f  fragments n = %s" % (label, ts[2], ts[0], e, [f.nao for f in be.Fobjs]), file=sys.stderr, flush=True)
