"""octane/STO-3G BE2 density matching with relax_density=True: iteration count and time."""
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mol = Mole(G / "octane.xyz")
mf = RHF(mol); mf.kernel()
be = BE(mf, FragPart.from_json(G / "fragmentation.json", "test_autogen_octane_be2"), distribute=False)
for relax in (False, True):
    t = time.time()
    out = be._sweep(None, eeval=True, return_vec=True, relax_density=relax)
    print("sweep relax=%s  %.3f s  err %.3e  E %.10f" % (relax, time.time() - t, out[0], out[2][0]), flush=True)
from quemb_amd.fragsolver import default_opts
be.opts = be.opts or default_opts()
be.opts.lambda_conv_tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
if len(sys.argv) > 2:
    be.opts.cc_conv_tol_normt = float(sys.argv[2]); be.opts.cc_conv_tol = float(sys.argv[2]) * 1e-2
t = time.time()
opt = be.optimize(solver="CCSD", only_chem=False, relax_density=True)
print("optimize relaxed: %.2f s, iterations %d, objfunc calls %d, err %.3e, E_corr %.10f" % (time.time() - t, be.beopt.iter, be.beopt.n_objfunc, be.beopt.err, be.e_corr), flush=True)
