"""Does the row stride of C matter for the K = n products of the MO transformation?  The batched C^T . slab product (M = N = K = 220, one n x n slab per
pair, 224 x 128 tile) and the flat slab . C product (M = npair * n, N = K = 220, 128 x 224 tile) with ldc = 220 (rows of 1760 bytes: every row segment
of a tile ends in a partially written 128-byte line) and ldc = 224."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
n = 220
npair = n * (n + 1) // 2
rng = np.random.default_rng(0)
dC = DeviceBuffer.from_numpy(rng.standard_normal((n, n)))
blk = DeviceBuffer.from_numpy(rng.standard_normal(1 << 22))


def rand_dev(nelem):
    d = DeviceBuffer(nelem); off = 0
    while off < nelem:
        m = min(1 << 22, nelem - off); check(lib.qemb_d2d(d.at(off), blk.ptr, m * 8)); off += m
    return d


def timed(f, reps=5):
    f(); lib.qemb_sync(); lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); f(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64(); lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value


flop = 2.0 * n * n * n * npair
for lds in (220, 224):          # row stride of the slab operand
    X = rand_dev(npair * n * lds)
    for ldc in (220, 224):
        Y = DeviceBuffer(npair * n * ldc)
        lib.qemb_set_gemm_config(13)
        t = timed(lambda: check(lib.qemb_op_gemm(n, n, n, 1.0, dC.ptr, n, 0, 0, X.ptr, lds, 0, n * lds, 0.0, Y.ptr, ldc, n * ldc, npair)))
        print(json.dumps(dict(product="batched C^T . slab (224 x 128 tile)", ld_slab=lds, ldc=ldc, ms=round(t, 3), tflops=round(flop / t / 1e9, 2))), flush=True)
        lib.qemb_set_gemm_config(34)
        t = timed(lambda: check(lib.qemb_op_gemm(npair * n, n, n, 1.0, X.ptr, lds, 1, 0, dC.ptr, n, 0, 0, 0.0, Y.ptr, ldc, 0, 1)))
        print(json.dumps(dict(product="flat slab . C (128 x 224 tile)", ld_slab=lds, ldc=ldc, ms=round(t, 3), tflops=round(flop / t / 1e9, 2))), flush=True)
        lib.qemb_set_gemm_config(-1)
        Y.free()
    X.free()
