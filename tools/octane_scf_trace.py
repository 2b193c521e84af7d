"""Host-side step times of the fragment RHF phase (QEMB_SCF_TRACE=1): one octane BE2 fragment alone, then the six of a lock-step sweep side by side."""
import os, sys
os.environ["QEMB_SCF_TRACE"] = "1"
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
for kw in (dict(nstreams=1), dict(lockstep=True)):
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False, **kw)
    be.oneshot(); be.oneshot()
    print("==== timed sweep", kw, file=sys.stderr, flush=True)
    be.oneshot()
