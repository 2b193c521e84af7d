"""List every GEMM of ONE converged fragment solve with its own device time (QEMB_GEMM_TRACE=1): which products an amplitude update is made of."""
import os, sys
os.environ["QEMB_GEMM_TRACE"] = "1"
os.environ["QEMB_GRAPH"] = "0"
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from quemb_amd import _lib
from quemb_amd.fragsolver import DeviceFragment, default_opts
from frag_bench import synthetic_on_device

lib = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
o = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h, d4, _dB, _naux = synthetic_on_device(lib, n, 20260803, scale=0.03); _dB.free()
fr = DeviceFragment(n, max(1, n // 10), lib=lib)
fr.set_eri_s4_dev(d4.ptr)
out = fr.solve(o, h, opts=default_opts(lib, cc_max_cycle=3), eeval=False)
lib.qemb_sync()
print("iterations", out["n_iter"], file=sys.stderr)
