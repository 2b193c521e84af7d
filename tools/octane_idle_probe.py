"""Is the one-off slow sweep after a new BE object an idle / power-state effect?  One lock-step BE object; sweeps with pauses of
different lengths in between (host sleeps, device idle), then a second BE object constructed and dropped in the middle of the series."""
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
mk = lambda: BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False, lib=lib, lockstep=True)
be = mk()


def sweeps(n, tag):
    ts = []
    for _ in range(n):
        lib.qemb_device_sync(); t0 = time.perf_counter(); be.oneshot(); lib.qemb_device_sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("RESULT %-44s %s" % (tag, " ".join("%.1f" % t for t in ts)), file=sys.stderr, flush=True)


sweeps(6, "cold start")
for pause in (0.05, 0.2, 1.0, 3.0):
    time.sleep(pause)
    sweeps(4, "after %.2f s of idle" % pause)
t0 = time.perf_counter(); other = mk(); dt = time.perf_counter() - t0
sweeps(4, "after constructing another BE (%.0f ms)" % (dt * 1e3))
del other
sweeps(4, "after dropping it")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:          # host busy (numpy), device idle
    import numpy as np; np.linalg.eigh(np.random.rand(200, 200))
sweeps(4, "after 0.3 s of host-only work")
