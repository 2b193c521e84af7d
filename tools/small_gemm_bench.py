"""GPU: the square products of small and mid-size fragments (ring products, M = N = K = n_occ n_virt) as a batch of `peers` products -- what a grouped
launch of a lock-step sweep runs -- on the candidate tiles.

    python tools/small_gemm_bench.py [cfg ...]

Round 5 (k-steps of 32 / 64 on the 64 x 64 tile and of 64 on the 32 x 32 tile, built for this measurement and removed again): no variant beat the 32 x 32 /
64 x 64 tiles with k-steps of 32 / 16 -- 27-29 TFLOP/s for six products of 440^3, 37-40 for four of 656^3 or 784^3, 41 / 54 for one / four of 1440^3; k-steps of 64
fell to 7-10 (one workgroup per CU).  These products are not waiting for one long k-step latency.
"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
cfgs = [int(a) for a in sys.argv[1:]] or [2, 1, 37]
rng = np.random.default_rng(0)
for (nov, peers, what) in ((440, 6, "octane BE2: six fragments, o v ~ 440"), (656, 4, "octane BE3: four fragments, o v ~ 650"), (656, 2, "BE3, two of the four (one launch per load width)"),
                           (784, 4, "n = 96: four fragments"), (1440, 1, "n = 132: one fragment"), (1440, 4, "n = 132: four fragments")):
    dA, dB, dC = (DeviceBuffer.from_numpy(rng.standard_normal(peers * nov * nov)) for _ in range(3))
    for cfg in cfgs:
        lib.qemb_set_gemm_config(cfg)

        def run():
            check(lib.qemb_op_gemm(nov, nov, nov, 1.0, dA.ptr, nov, 1, nov * nov, dB.ptr, nov, 1, nov * nov, 0.0, dC.ptr, nov, nov * nov, peers))
        try:
            run(); run(); lib.qemb_sync()
        except Exception as e:  # noqa: BLE001
            print(json.dumps(dict(nov=nov, peers=peers, cfg=cfg, failed=str(e)[:100])), flush=True)
            continue
        lib.qemb_timer_reset(5)
        for _ in range(20):
            lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
        ms, cnt = C.c_double(), C.c_int64()
        lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
        t = ms.value / cnt.value
        print(json.dumps(dict(what=what, nov=nov, peers=peers, cfg=cfg, us=round(t * 1e3, 1), tflops=round(2.0 * peers * nov ** 3 / t / 1e9, 1))), flush=True)
    lib.qemb_set_gemm_config(-1)
    for b in (dA, dB, dC):
        b.free()
