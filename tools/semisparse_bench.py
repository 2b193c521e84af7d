"""Semi-sparse DF transform at scale: N AOs with a banded exch_reachable (each AO reaches `band` neighbours on
either side), naux auxiliary functions, n embedding orbitals localised on a stretch of the chain.  Reports device memory held by the
tensor (O(n_unique naux)) against the dense (P|mu nu), and the transform time with and without MO screening."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd import eri_transform as et

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
band = int(sys.argv[2]) if len(sys.argv) > 2 else 60
naux = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
lib = _lib.init(0)
rng = np.random.default_rng(3)
reach = [list(range(max(0, mu - band), min(N, mu + band + 1))) for mu in range(N)]
t = et.SemiSparseSym3DTensor((naux, N, N), reach)
t.unique_dense_data[:] = rng.standard_normal(t.unique_dense_data.shape) * 0.05
# embedding orbitals: orthonormal, localised on AOs [N/2 - 2n, N/2 + 2n)
TA = np.zeros((N, n))
lo, hi = max(0, N // 2 - 2 * n), min(N, N // 2 + 2 * n)
TA[lo:hi] = np.linalg.qr(rng.standard_normal((hi - lo, n)))[0]
S_abs = np.exp(-0.15 * np.abs(np.subtract.outer(np.arange(N), np.arange(N))))
A = rng.standard_normal((naux, naux)) * 0.01
Lc = np.linalg.cholesky(A @ A.T + np.eye(naux))
df = et.DFContext(L_PQ=Lc, lib=lib)
t0 = time.time(); df.set_ints_semisparse(t); lib.qemb_sync(); t_up = time.time() - t0
res = dict(N=N, band=band, naux=naux, n=n, n_unique=t.unique_dense_data.shape[1], tensor_GB=t.unique_dense_data.nbytes / 1e9,
           dense_GB=naux * N * N * 8 / 1e9, upload_s=t_up)
from quemb_amd.fragsolver import DeviceFragment
fr = DeviceFragment(n, 1, lib=lib)                             # the fragment ERIs stay on the device, as in BE
for label, kw in (("no_mo_screening", {}), ("eps_1e-5", dict(S_abs=S_abs, MO_coeff_epsilon=1e-5))):
    df.transform(TA, frag=fr, want_host=False, **kw)           # warm-up (workspaces)
    lib.qemb_sync()
    t0 = time.time(); df.transform(TA, frag=fr, want_host=False, **kw); lib.qemb_sync(); res[label + "_s"] = time.time() - t0
out = df.transform(TA, want_host=True, S_abs=S_abs, MO_coeff_epsilon=1e-5)
res["checksum"] = float(np.abs(out).sum())
# first-contraction flops: 2 * sum_mu |reach(mu)| * n * naux
res["first_contraction_GF"] = 2.0 * sum(len(r) for r in reach) * n * naux / 1e9
print(json.dumps(res))
