"""octane/STO-3G autogen BE3 density matching (reference golden for chemgen BE3: E_corr = -0.5497021857717073, tests/molbe_octane_test.py:63-68)."""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
t = time.time()
be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be3"), distribute=False)
print("init %.2f s; fragments n = %s" % (time.time() - t, [f.nao for f in be.Fobjs]), flush=True)
t = time.time()
e, _ = be.oneshot()
print("one-shot %.3f s  E_corr %.10f" % (time.time() - t, e), flush=True)
t = time.time()
be.optimize(solver="CCSD", only_chem=False)
print("matching %.2f s  iterations %d  err %.2e  E_corr %.10f  E_tot %.10f  (golden chemgen BE3 -0.5497021857717073 / -310.3344717358742)" %
      (time.time() - t, be.beopt.iter, be.beopt.err, be.e_corr, be.ebe_tot), flush=True)
