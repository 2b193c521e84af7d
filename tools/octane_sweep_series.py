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
for kw in (dict(nstreams=6, lockstep=False), dict(lockstep=True)):
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False, lib=lib, **kw)
    be.oneshot()
    ts = []
    for _ in range(25):
        lib.qemb_device_sync(); t0 = time.perf_counter()
        be.oneshot()
        lib.qemb_device_sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("RESULT", kw, " ".join("%.1f" % t for t in ts), file=sys.stderr, flush=True)
