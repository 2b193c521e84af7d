"""GPU micro-benchmark: J/K build of the fragment RHF at n = 220 -- one pass over the packed block (dev_jk_from_packed) against the
Coulomb product over the packed block plus the exchange build over the pair-row tensor (the two passes it replaced)."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
rng = np.random.default_rng(0)
npair = n * (n + 1) // 2
Bp = rng.standard_normal((8, npair))
dB, d4 = DeviceBuffer.from_numpy(Bp), DeviceBuffer(npair * npair)
check(lib.qemb_op_gemm(npair, npair, 8, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
D = rng.standard_normal((n, n)); D = D + D.T
il = np.tril_indices(n)
Dp = np.ascontiguousarray((2 * D - np.diag(np.diag(D)))[il])
dD, dDp, dJ, dK = DeviceBuffer.from_numpy(D), DeviceBuffer.from_numpy(Dp), DeviceBuffer(npair), DeviceBuffer(n * n)
dH = DeviceBuffer(npair * n * n)
check(lib.qemb_op_unpack_tril_rows(npair, n, d4.ptr, dH.ptr))


def timed(f, reps=10):
    f(); f(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); f(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value


t_new = timed(lambda: check(lib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, dDp.ptr, dJ.ptr, dK.ptr)))
t_k = timed(lambda: check(lib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, None, None, dK.ptr)))
t_old_k = timed(lambda: check(lib.qemb_op_k_from_pairs(n, dH.ptr, dD.ptr, dK.ptr)))
t_old_j = timed(lambda: check(lib.qemb_op_gemv_rows(npair, npair, d4.ptr, npair, dDp.ptr, dJ.ptr, 1.0, 0.0)))
gb = npair * npair * 8 / 1e9
print(json.dumps(dict(n=n, packed_block_GB=gb, jk_one_pass_ms=t_new, k_only_one_pass_ms=t_k, old_k_pair_rows_ms=t_old_k, old_j_gemv_ms=t_old_j,
                      one_pass_TBps=gb / t_new, old_k_TBps=2 * gb / t_old_k, old_j_TBps=gb / t_old_j)))
