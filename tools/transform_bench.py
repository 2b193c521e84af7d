"""AO -> fragment ERI transforms at the SURVEY 8(d) sizes (rows a3/a4): dense N_ao = 256 (s8, 4.3 GB) and DF N_ao = 512,
naux = 1000, fragment n = 220.  ERIs land in a device-resident fragment (no host copy in the timed region)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib, eri_transform as et
from quemb_amd.fragsolver import DeviceFragment

lib = _lib.init(0)
rng = np.random.default_rng(20260803)
n = 220


def timed(fn, reps=3):
    fn(); lib.qemb_sync()
    ts = []
    for _ in range(reps):
        t = time.time(); fn(); lib.qemb_sync(); ts.append(time.time() - t)
    return min(ts)


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("dense", "both"):
    N = 256
    npair = N * (N + 1) // 2
    B = 0.06 * rng.standard_normal((64, npair))
    s4 = B.T @ B
    s8 = s4[np.tril_indices(npair)]
    del s4
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n].copy()
    t = time.time(); ao = et.AOEri(s8, N, lib=lib); lib.qemb_sync(); t_up = time.time() - t
    fr = DeviceFragment(n, 22, lib=lib)
    dt = timed(lambda: ao.transform(TA, frag=fr, want_host=False))
    flop = 2.0 * N ** 4 * n + 2.0 * N ** 3 * n ** 2 + 2.0 * N ** 2 * n ** 3 + 2.0 * N * n ** 4
    # spot check against the DF factor: (ij|kl) = sum_P b_P,ij b_P,kl with b = TA^T B TA
    out = ao.transform(TA, want_host=True)
    il = np.tril_indices(N)
    Bf = np.zeros((64, N, N)); Bf[:, il[0], il[1]] = B; Bf = Bf + Bf.transpose(0, 2, 1); Bf[:, np.arange(N), np.arange(N)] *= 0.5
    b = np.einsum("Ppq,pi,qj->Pij", Bf, TA, TA, optimize=True)
    iln = np.tril_indices(n)
    bp = b[:, iln[0], iln[1]]
    err = float(np.abs(out - bp.T @ bp).max())
    print(json.dumps(dict(case="dense a3", N_ao=N, n=n, s8_GB=s8.nbytes / 1e9, upload_s=t_up, transform_ms=dt * 1e3,
                          full_flop=flop, tflops_full_equiv=flop / dt / 1e12, max_abs_err=err)), flush=True)
    ao.free(); del fr
if which in ("df", "both"):
    N, naux = 512, 1000
    npair = N * (N + 1) // 2
    ints = 0.06 * rng.standard_normal((naux, npair))
    A = rng.standard_normal((naux, naux)) / np.sqrt(naux)
    j2c = A @ A.T + np.eye(naux)
    TA = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :n].copy()
    t = time.time(); df = et.DFContext(j2c=j2c, lib=lib); df.set_ints(ints, N, layout="packed"); lib.qemb_sync(); t_up = time.time() - t
    fr = DeviceFragment(n, 22, lib=lib)
    dt = timed(lambda: df.transform(TA, frag=fr, want_host=False))
    npn = n * (n + 1) // 2
    # EXECUTED flops (ao2mo.cpp DfContext::transform): the two rotations, the product with the dense L^-1 (a full GEMM, twice the TRSM count),
    # and the block columns of bb^T bb at and below the diagonal (8 column blocks, the rest is mirrored)
    nblk = 8 if npn >= 2048 else 1
    w = ((npn + nblk - 1) // nblk + 127) // 128 * 128
    syrk = sum(2.0 * (npn - c0) * min(w, npn - c0) * naux for c0 in range(0, npn, w))
    flop = 2.0 * naux * N * N * n + 2.0 * naux * N * n * n + 2.0 * naux * naux * npn + syrk
    flop_unsym = 2.0 * naux * N * N * n + 2.0 * naux * N * n * n + 1.0 * naux * naux * npn + 2.0 * naux * npn * npn   # TRSM + full bb^T bb
    out = df.transform(TA, want_host=True)
    il = np.tril_indices(N)
    Bf = np.zeros((naux, N, N)); Bf[:, il[0], il[1]] = ints; Bf = Bf + Bf.transpose(0, 2, 1); Bf[:, np.arange(N), np.arange(N)] *= 0.5
    b = np.einsum("Ppq,pi,qj->Pij", Bf, TA, TA, optimize=True)
    iln = np.tril_indices(n)
    bp = np.linalg.solve(np.linalg.cholesky(j2c), b[:, iln[0], iln[1]])
    err = float(np.abs(out - bp.T @ bp).max())
    print(json.dumps(dict(case="DF a4", N_ao=N, naux=naux, n=n, setup_s=t_up, transform_ms=dt * 1e3, flop_executed=flop,
                          tflops_executed=flop / dt / 1e12, flop_unsymmetrised=flop_unsym, tflops_full_equiv=flop_unsym / dt / 1e12,
                          max_abs_err=err)), flush=True)
