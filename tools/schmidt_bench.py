"""Schmidt decomposition timing at the SURVEY 8(d) size: C = QR(N(0,1)) with N_lo = 2048, nocc = N_lo/4, 22-site fragment."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib, eri_transform as et
lib = _lib.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(20260803)
C = np.linalg.qr(rng.standard_normal((N, N)))[0]
frag = list(range(100, 122))
for method in ("subspace", "eigh"):
    et.schmidt_decomposition(C[:200, :200].copy(), 50, list(range(10)), method=method)   # warm-up
    t = time.time(); TA, nf, nb = et.schmidt_decomposition(C, N // 4, frag, method=method); dt = time.time() - t
    D = C[:, : N // 4] @ C[:, : N // 4].T
    print(json.dumps(dict(method=method, N_lo=N, nocc=N // 4, n_f=nf, n_b=nb, wall_s=dt, orth_err=float(np.abs(TA.T @ TA - np.eye(nf + nb)).max()),
                          electrons_in_embedding=float(np.trace(TA.T @ D @ TA)))), flush=True)
t = time.time(); w, v = np.linalg.eigh(D[np.ix_([i for i in range(N) if i not in frag], [i for i in range(N) if i not in frag])]); print(json.dumps(dict(method="numpy eigh (LAPACK, host)", wall_s=time.time() - t)))
