"""Per-phase device time of one octane BE2 sweep (small, launch-bound fragments)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd import _lib
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
lib = _lib.init(0)
mol = Mole(G / "octane.xyz"); mf = RHF(mol); mf.kernel()
be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False)
be.oneshot()
for s in range(8): lib.qemb_timer_reset(s)
t = time.time(); be.oneshot(); wall = time.time() - t
out = {}
for name, slot in dict(ladder=0, rings=1, iter=2, ao2mo=3, scf=4).items():
    ms = C.c_double(); cnt = C.c_int64(); lib.qemb_timer_read(slot, C.byref(ms), C.byref(cnt)); out[name] = (round(ms.value, 2), cnt.value)
print("wall ms", wall * 1e3, out)
