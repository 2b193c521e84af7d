import sys, time, os
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
for key in ("test_autogen_octane_be2",):
    for ns in (1, 6, "lockstep"):
        kw = dict(lockstep=True) if ns == "lockstep" else dict(nstreams=ns)
        be = BE(mf, FragPart.from_json(G / "fragmentation.json", key), distribute=False, **kw)
        be.oneshot(); be.oneshot()
        t = time.time()
        for _ in range(5):
            e, _ = be.oneshot()
        dt = (time.time() - t) / 5
        print("RESULT %s nstreams=%s sweep %.1f ms E_corr %.12f hwq=%s graph=%s" % (key, ns, dt * 1e3, e, os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("QEMB_GRAPH")), be.stats if ns == "lockstep" else "", flush=True)
