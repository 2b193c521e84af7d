"""Tile / K-split choice for the Fvv' product of the CCSD update (M = N = n_virt, K = n_occ^2 n_virt, both operands stored [K][M]): a 200 x 200 result from
two 128 MB operands -- an HBM pass, not an MFMA problem.  Prints the time per (tile config, K split)."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
o, v = 20, 200
M = N = v
K = o * o * v
rng = np.random.default_rng(0)


def rand_dev(nelem):
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


dA, dB, dC = rand_dev(K * M), rand_dev(K * N), DeviceBuffer(M * N)
ref = None
for cfg, ks in ((1, 0), (1, 96), (1, 192), (0, 0), (0, 96), (0, 192), (0, 312), (4, 192), (4, 312), (13, 192), (13, 312), (34, 192), (34, 312), (2, 0)):
    lib.qemb_set_gemm_config(cfg)
    lib.qemb_set_gemm_ksplit(ks)

    def run():
        check(lib.qemb_op_gemm(M, N, K, -1.0, dA.ptr, M, 0, 0, dB.ptr, N, 0, 0, 0.0, dC.ptr, N, 0, 1))
    run(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(10):
        lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    got = dC.numpy((M, N))
    if ref is None:
        ref = got.copy()
    t = ms.value / cnt.value
    print(json.dumps(dict(cfg=cfg, ksplit=ks, us=round(t * 1e3, 1), TBps=round(2 * K * M * 8 / t / 1e9, 2), max_dev_from_first=float(np.abs(got - ref).max()))), flush=True)
lib.qemb_set_gemm_config(-1); lib.qemb_set_gemm_ksplit(0)
