"""octane BE2 sweeps through the lock-step batch path only (profiling aid): python tools/octane_lockstep.py [key] [sweeps]"""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
key = sys.argv[1] if len(sys.argv) > 1 else "test_autogen_octane_be2"
nsweep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
be = BE(mf, FragPart.from_json(G / "fragmentation.json", key), distribute=False, lockstep=True)
be.oneshot()
t = time.time()
for _ in range(nsweep):
    e, _ = be.oneshot()
print("RESULT %s lockstep sweep %.1f ms E_corr %.12f" % (key, (time.time() - t) / nsweep * 1e3, e), be.stats, flush=True)
