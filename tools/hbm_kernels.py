"""GPU micro-benchmark of the HBM-bound kernels of one fragment solve at the benchmarked size (n = 220, n_occ = 20, n_virt = 200), each on its
own with the operand shapes of the solve: algorithmic bytes (every operand read or written once) / HIP-event time, against 8 TB/s.

    python tools/hbm_kernels.py [name ...]          # default: all;   QEMB_JK_V1=1 ... : the round-2 J/K kernel (A/B)
"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
import os
n, o = int(os.environ.get("QEMB_HBM_N", "220")), 20      # QEMB_HBM_N: another fragment size (alignment experiments)
v = n - o
npn, npo, nmo, npv, nmv, nov = n * (n + 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2, o * v
ldp, ldm = npv + (npv & 1), nmv + (nmv & 1)
rng = np.random.default_rng(0)
want = set(sys.argv[1:])
print(json.dumps(dict(n=n, n_occ=o)), flush=True)


def timed(f, reps=10):
    f(); f(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); f(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value


def report(name, what, nbytes, ms):
    print(json.dumps(dict(kernel=name, what=what, algorithmic_GB=round(nbytes / 1e9, 3), ms=round(ms, 4), TBps=round(nbytes / ms / 1e9, 2),
                          frac_of_8_TBps=round(nbytes / ms / 1e9 / 8.0, 3))), flush=True)


def rand_dev(nelem):
    """a device buffer of pseudo-random numbers without a multi-GB host array: a small block tiled by device copies"""
    blk = rng.standard_normal(min(nelem, 1 << 22))
    d = DeviceBuffer(nelem)
    src = DeviceBuffer.from_numpy(blk)
    off = 0
    while off < nelem:
        m = min(blk.size, nelem - off)
        check(lib.qemb_d2d(d.at(off), src.ptr, m * 8))
        off += m
    src.free()
    return d


if not want or "jk" in want:
    d4 = rand_dev(npn * npn)
    D = rng.standard_normal((n, n)); D = D + D.T
    il = np.tril_indices(n)
    Dp = np.ascontiguousarray((2 * D - np.diag(np.diag(D)))[il])
    dD, dDp, dJ, dK = DeviceBuffer.from_numpy(D), DeviceBuffer.from_numpy(Dp), DeviceBuffer(npn), DeviceBuffer(n * n)
    t = timed(lambda: check(lib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, dDp.ptr, dJ.ptr, dK.ptr)))
    report("jk_from_packed", "J and K from the 4-fold packed block, one pass (fragment RHF, 5 builds per solve)", npn * npn * 8, t)
    t = timed(lambda: check(lib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, None, None, dK.ptr)))
    report("jk_from_packed (K only)", "exchange matrix only", npn * npn * 8, t)
    dH = DeviceBuffer(npn * n * n)
    t = timed(lambda: check(lib.qemb_op_unpack_tril_rows(npn, n, d4.ptr, dH.ptr)), reps=5)
    report("unpack_tril_rows", "s4 block -> [pq][r][s] (read packed + write unpacked)", (npn * npn + npn * n * n) * 8, t)
    dX = DeviceBuffer(npn * n * n)
    t = timed(lambda: check(lib.qemb_op_unpack_tril_pair_rows(n, n, dH.ptr, dX.ptr)), reps=5)
    report("unpack_tril_pair_rows", "[r'][s'][pq] rows r' >= s' -> [(r's')][p][q]", (npn * npn + npn * n * n) * 8, t)
    for b in (d4, dD, dDp, dJ, dK, dH, dX):
        b.free()
    lib.qemb_trim()

if not want or "pack" in want:
    dIn = rand_dev(nov * v * v)
    dOp, dOm = DeviceBuffer(nov * ldp), DeviceBuffer(nov * ldm)
    t = timed(lambda: check(lib.qemb_op_pack_pm_cols(nov, v, dIn.ptr, dOp.ptr, ldp, dOm.ptr, ldm)))
    report("pack_pm_cols", "(+/-) packed images of the ovvv block: o v slabs of v x v", (nov * v * v + nov * (npv + nmv)) * 8, t)
    dTau = rand_dev(o * o * v * v)
    dTp, dTm = DeviceBuffer(npo * ldp), DeviceBuffer(max(nmo, 1) * ldm)
    t = timed(lambda: check(lib.qemb_op_ladder_pack_tau(o, v, dTau.ptr, dTp.ptr, ldp, dTm.ptr, ldm)), reps=30)
    report("ladder_pack_tau", "tau -> (+/-) packed pair rows (every iteration)", (npo * v * v + npo * npv + nmo * nmv) * 8, t)
    for b in (dIn, dOp, dOm, dTau, dTp, dTm):
        b.free()

if not want or "smallk" in want:
    oo, vv = o * o, v * v
    dC = rand_dev(oo * vv)
    dA = rand_dev(oo * o * v)
    dt1 = DeviceBuffer.from_numpy(rng.standard_normal((o, v)))
    t = timed(lambda: check(lib.qemb_op_small_k_update(oo, v, v, o, -1.0, dA.ptr, nov, dt1.ptr, 0, dC.ptr, vv)), reps=30)
    report("small_k_update (U)", "U[ij][a][b] -= sum_k X[ij][k][a] t1[k][b]: r/w o^2 v^2", (2 * oo * vv + oo * o * v) * 8, t)
    dZ = rand_dev(nov * v * o)
    dB = rand_dev(nov * oo)
    t = timed(lambda: check(lib.qemb_op_small_k_update(nov, v, o, o, -1.0, dt1.ptr, 0, dB.ptr, oo, dZ.ptr, v * o)), reps=30)
    report("small_k_update (ZB)", "ZB[k,c,a,i] -= sum_l t1[l,a] ovoo[k,c,l,i]: r/w o^2 v^2, N = n_occ columns", (2 * oo * vv + nov * oo) * 8, t)
    for b in (dC, dA, dt1, dZ, dB):
        b.free()

if not want or "layouts" in want:
    oo, vv = o * o, v * v
    N2 = oo * vv
    d2 = rand_dev(N2)
    dt1 = DeviceBuffer.from_numpy(rng.standard_normal((o, v)))
    outs = [DeviceBuffer(N2) for _ in range(6)]
    t = timed(lambda: check(lib.qemb_op_ccsd_ph_layouts(o, v, d2.ptr, dt1.ptr, *[b.ptr for b in outs])), reps=30)
    report("ccsd_ph_layouts", "t2 -> T, T', u, u~, T'~, Theta in one pass (7 o^2 v^2)", 7 * N2 * 8, t)
    for b in [d2, dt1] + outs:
        b.free()

if not want or "copy4" in want:
    oo, vv = o * o, v * v
    N2 = oo * vv
    L = C.c_int64 * 4
    dZB, dU = rand_dev(N2), rand_dev(N2)
    # U[i,j,a,b] += ZB[i,a,b,j]: dims (i, j, a, b), input strides of ZB[i][a][b][j], output strides of U
    t = timed(lambda: check(lib.qemb_op_copy4(L(o, o, v, v), dZB.ptr, L(v * v * o, 1, v * o, o), dU.ptr, L(o * vv, vv, v, 1), 1.0, 1.0)), reps=30)
    report("copy4 (U += ZB)", "U[i,j,a,b] += ZB[i,a,b,j]: the n_occ-long index is the contiguous one of the input (3 o^2 v^2)", 3 * N2 * 8, t)
    # W1[i,a,k,c] = ZB[k,c,a,i] (+ base in the solve): dims (i, a, k, c)
    t = timed(lambda: check(lib.qemb_op_copy4(L(o, v, o, v), dZB.ptr, L(1, o, vv * o, v * o), dU.ptr, L(v * o * v, o * v, v, 1), 1.0, 1.0)), reps=30)
    report("copy4 (W1 += ZB)", "W1[i,a,k,c] += ZB[k,c,a,i] (3 o^2 v^2)", 3 * N2 * 8, t)
    for b in (dZB, dU):
        b.free()
