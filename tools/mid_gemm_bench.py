"""GPU: the two product shapes that carry a mid-size CCSD iteration, on the candidate tiles and K splits.

    python tools/mid_gemm_bench.py

(a) ring products, M = N = K = n_occ n_virt, ONE fragment (single stream): tiles 32 x 32 / 64 x 64 / 96 x 96 x explicit K splits;
(b) pair-row products of the pp-ladder, M = npair(n_occ), N = K = npair(n_virt): row tiles 48 / 64 / 80 x K splits.
"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
rng = np.random.default_rng(0)


def timed(run, reps=20):
    run(); run(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
    ms, cnt = C.c_double(), C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value


for nov in (348, 783, 1440, 2560):
    dA, dB, dC = (DeviceBuffer.from_numpy(rng.standard_normal(nov * nov)) for _ in range(3))
    for cfg in (2, 1, 37):
        for ks in (0, 2, 3, 4):
            lib.qemb_set_gemm_config(cfg); lib.qemb_set_gemm_ksplit(ks)
            try:
                t = timed(lambda: check(lib.qemb_op_gemm(nov, nov, nov, 1.0, dA.ptr, nov, 1, 0, dB.ptr, nov, 1, 0, 0.0, dC.ptr, nov, 0, 1)))
            except Exception as e:  # noqa: BLE001
                print(json.dumps(dict(shape="ring", nov=nov, cfg=cfg, ks=ks, failed=str(e)[:80])), flush=True); continue
            print(json.dumps(dict(shape="ring", nov=nov, cfg=cfg, ks=ks, us=round(t * 1e3, 1), tflops=round(2.0 * nov ** 3 / t / 1e9, 1))), flush=True)
    for b in (dA, dB, dC):
        b.free()
for (o, v) in ((6, 58), (9, 87), (12, 120)):
    M, N = o * (o + 1) // 2, v * (v + 1) // 2
    ld = N + (N & 1)
    dA = DeviceBuffer.from_numpy(rng.standard_normal(M * ld)); dB = DeviceBuffer.from_numpy(rng.standard_normal(N * ld)); dC = DeviceBuffer.from_numpy(rng.standard_normal(8 * M * ld))
    for cfg in (38, 12, 36, 21, 1):
        for ks in (0, 4, 8, 16):
            lib.qemb_set_gemm_config(cfg); lib.qemb_set_gemm_ksplit(ks)
            try:
                t = timed(lambda: check(lib.qemb_op_gemm(M, N, ld, 1.0, dA.ptr, ld, 1, 0, dB.ptr, ld, 1, 0, 0.0, dC.ptr, ld, 0, 1)))
            except Exception as e:  # noqa: BLE001
                print(json.dumps(dict(shape="pair rows", o=o, v=v, cfg=cfg, ks=ks, failed=str(e)[:80])), flush=True); continue
            print(json.dumps(dict(shape="pair rows", o=o, v=v, M=M, N=N, cfg=cfg, ks=ks, us=round(t * 1e3, 1), tflops=round(2.0 * M * N * ld / t / 1e9, 1))), flush=True)
    for b in (dA, dB, dC):
        b.free()
lib.qemb_set_gemm_config(-1); lib.qemb_set_gemm_ksplit(0)
