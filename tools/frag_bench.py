"""Time one synthetic fragment (SURVEY 8d family) through the device pipeline; print per-phase device times."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check
from quemb_amd.fragsolver import DeviceFragment, default_opts


def synthetic_on_device(lib, n, seed, naux=None, scale=None, gap=2.0):
    """h (host) and the s4-packed ERIs built ON THE DEVICE from the DF factor (B^T B over packed pairs)."""
    rng = np.random.default_rng(seed)
    naux = naux or 3 * n
    scale = 0.06 * min(1.0, (55.0 / n) ** 0.5) if scale is None else scale
    B = scale * rng.standard_normal((naux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])          # (naux, npair)
    npair = Bp.shape[1]
    dB = DeviceBuffer.from_numpy(Bp)
    d4 = DeviceBuffer(npair * npair)
    check(lib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
    A = rng.standard_normal((n, n))
    h = np.diag(gap * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    return h, d4, dB, naux


def timers(lib):
    out = {}
    for name, slot in dict(ladder=0, rings=1, iter=2, ao2mo=3, scf=4).items():
        ms = C.c_double(); cnt = C.c_int64()
        lib.qemb_timer_read(slot, C.byref(ms), C.byref(cnt))
        out[name] = dict(ms=ms.value, count=cnt.value)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
    o = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lib = _lib.init(0)
    t0 = time.time()
    h, d4, dB, naux = synthetic_on_device(lib, n, 20260803)
    fr = DeviceFragment(n, min(22, n // 2))
    # "four-index" among the arguments: the four quarter transformations of the packed block instead of the 3-index factor route; "eeval": with the energies' 3/4 blocks;
    # "block": keep the 4-fold packed block resident beside the factor (rounds 1-4) -- the default since round 5 is bench.py's: the fragment lives on its factor alone
    four, eeval = "four-index" in sys.argv, "eeval" in sys.argv
    if four or "block" in sys.argv:
        fr.set_eri_s4_dev(d4.ptr)
        fr.set_df_factor_dev(dB.ptr, naux)
    else:
        fr.set_df_only_dev(dB.ptr, naux)
    d4.free(); dB.free()
    fr.set_mo_route(0 if four else -1)
    if eeval:
        rng = np.random.default_rng(1)
        V = rng.standard_normal((n, n))
        fr.set_energy_data(h, 0.05 * (V + V.T), None, 1.0, list(range(4)))
    lib.qemb_sync(); print("setup s", time.time() - t0, flush=True)
    for s in range(8): lib.qemb_timer_reset(s)
    t0 = time.time()
    out = fr.solve(o, h, opts=default_opts(verbose=int("verbose" in sys.argv)), eeval=eeval)
    lib.qemb_sync(); print("first solve wall s", time.time() - t0, flush=True)
    for s in range(8): lib.qemb_timer_reset(s)
    t0 = time.time()
    # as in a BE sweep: the one-body matrix moved a little (a new effective potential), dm0 is the previous solve's density
    Cprev = out["mo_coeff"]
    dm0 = 2.0 * Cprev[:, :o] @ Cprev[:, :o].T
    h2 = h.copy(); h2[:4, :4] += 1e-3
    out = fr.solve(o, h2, dm0=dm0, opts=default_opts(verbose=0), eeval=eeval)
    lib.qemb_sync(); wall = time.time() - t0
    v = n - o
    tm = timers(lib)
    lad = tm["ladder"]["ms"] / max(tm["ladder"]["count"], 1)
    res = dict(n=n, o=o, mo_route="factor" if fr.mo_route_used()[0] else "four-index", resident_bytes=fr.resident_bytes(), eeval=eeval, wall_s=wall, n_iter=out["n_iter"], scf_cycles=out["scf_cycles"], e_corr=out["e_corr_mo"], timers=tm,
               ladder_ms=lad, ladder_tflops=2.0 * o * o * v ** 4 / (lad * 1e-3) / 1e12 if lad else None,
               iter_ms=tm["iter"]["ms"] / max(tm["iter"]["count"], 1))
    print(json.dumps(res), flush=True)
