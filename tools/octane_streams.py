"""Small-fragment regime in ONE process: octane BE2 / BE3 sweeps with 1..6 fragments in flight (solver.map_fragments: one host
thread + one execution context = HIP stream per fragment in flight)."""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
for key in ("test_autogen_octane_be2", "test_autogen_octane_be3"):
    for ns in (1, 2, 3, 6):
        be = BE(mf, FragPart.from_json(G / "fragmentation.json", key), distribute=False, nstreams=ns)
        be.oneshot(); be.oneshot()
        t = time.time()
        for _ in range(5):
            e, _ = be.oneshot()
        dt = (time.time() - t) / 5
        print("RESULT %s nstreams=%d sweep %.1f ms E_corr %.12f" % (key, ns, dt * 1e3, e), flush=True)
