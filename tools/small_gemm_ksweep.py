"""GPU: a batch of six 441 x 441 x K products (what a grouped launch of an octane lock-step sweep runs) for growing K on the 32 x 32 and 64 x 64 tiles: the time is
a + b K -- a = what a launch costs whatever its K (dispatch, prologue, epilogue, tail round), b = the rate of the main loop.   python tools/small_gemm_ksweep.py"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check
lib = _lib.init(0)
rng = np.random.default_rng(0)
M = N = 441; peers = 6
Ks = (32, 64, 128, 256, 441, 882, 1764, 3528)
Kmax = max(Ks)
dA, dB = (DeviceBuffer.from_numpy(rng.standard_normal(peers * M * Kmax)) for _ in range(2))
dC = DeviceBuffer(peers * M * N)
for cfg in (2, 1):
    lib.qemb_set_gemm_config(cfg)
    rows = []
    for K in Ks:
        run = lambda: check(lib.qemb_op_gemm(M, N, K, 1.0, dA.ptr, K, 1, M * K, dB.ptr, K, 1, N * K, 0.0, dC.ptr, N, M * N, peers))
        run(); run(); lib.qemb_sync()
        lib.qemb_timer_reset(5)
        for _ in range(20):
            lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
        ms, cnt = C.c_double(), C.c_int64()
        lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
        t = ms.value / cnt.value * 1e3
        rows.append((K, t))
        print(json.dumps(dict(cfg=cfg, K=K, us=round(t, 1), tflops=round(2.0 * peers * M * N * K / t / 1e6, 1))), flush=True)
    (k1, t1), (k2, t2) = rows[-3], rows[-1]
    b = (t2 - t1) / (k2 - k1)
    a = rows[4][1] - b * rows[4][0]
    wg = ((M + 31) // 32) ** 2 * peers if cfg == 2 else ((M + 63) // 64) ** 2 * peers
    print(json.dumps(dict(cfg=cfg, fit="t = a + b K from the two longest K", a_us=round(a, 1), b_us_per_k=round(b, 4), asymptotic_tflops=round(2.0 * peers * M * N / b / 1e6, 1),
                          at_K_441=dict(total_us=round(rows[4][1], 1), fixed_share=round(a / rows[4][1], 3)), workgroups=wg)), flush=True)
lib.qemb_set_gemm_config(-1)
