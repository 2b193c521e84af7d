"""GPU: time of a batch of `peers` square products (M = N = K = nov) on one tile configuration as the batch grows -- does the time grow with the work, or in steps
(rounds of resident workgroups)?   python tools/small_gemm_occupancy.py [nov] [cfg ...]"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check
lib = _lib.init(0)
nov = int(sys.argv[1]) if len(sys.argv) > 1 else 441
cfgs = [int(a) for a in sys.argv[2:]] or [2, 1]
rng = np.random.default_rng(0)
pmax = 16
dA, dB, dC = (DeviceBuffer.from_numpy(rng.standard_normal(pmax * nov * nov)) for _ in range(3))
for cfg in cfgs:
    lib.qemb_set_gemm_config(cfg)
    for peers in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        run = lambda: check(lib.qemb_op_gemm(nov, nov, nov, 1.0, dA.ptr, nov, 1, nov * nov, dB.ptr, nov, 1, nov * nov, 0.0, dC.ptr, nov, nov * nov, peers))
        run(); run(); lib.qemb_sync()
        lib.qemb_timer_reset(5)
        for _ in range(20):
            lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
        ms, cnt = C.c_double(), C.c_int64()
        lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
        t = ms.value / cnt.value
        print(json.dumps(dict(nov=nov, cfg=cfg, peers=peers, us=round(t * 1e3, 1), us_per_product=round(t * 1e3 / peers, 2), tflops=round(2.0 * peers * nov ** 3 / t / 1e9, 1))), flush=True)
lib.qemb_set_gemm_config(-1)
