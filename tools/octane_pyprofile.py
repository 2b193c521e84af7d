"""cProfile of the Python side of octane lock-step sweeps: python tools/octane_pyprofile.py [key] [sweeps]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
key = sys.argv[1] if len(sys.argv) > 1 else "test_autogen_octane_be2"
nsweep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
be = BE(mf, FragPart.from_json(G / "fragmentation.json", key), distribute=False, lockstep=True)
be.oneshot(); be.oneshot()
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
for _ in range(nsweep):
    be.oneshot()
pr.disable()
print("mean sweep %.2f ms (under cProfile)" % ((time.perf_counter() - t) / nsweep * 1e3), file=sys.stderr)
st = pstats.Stats(pr, stream=sys.stderr); st.sort_stats("cumulative").print_stats(35)
