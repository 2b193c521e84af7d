"""GPU micro-benchmark: tile configurations x main-loop variants of the FP64 MFMA GEMM on the shapes of one CCSD iteration at
n_occ = 20, n_virt = 200 (run via gpurun).  Prints one JSON line per (shape, cfg).

    python tools/gemm_modes.py [reps]
"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
PEAK = 78.6e12
rng = np.random.default_rng(0)


def mk(n):
    b = DeviceBuffer(n)
    chunk = rng.standard_normal(min(n, 1 << 24))
    off = 0
    while off < n:
        m = min(chunk.size, n - off)
        check(lib.qemb_h2d(b.at(off), chunk.ctypes.data, m * 8))
        off += m
    return b


def bench(tag, M, N, K, a_kc, b_kc, cfgs, ks=0, reps=6, ld=None):
    ldk = ld or K
    dA = mk(M * (ldk if a_kc else M) if a_kc else K * M)
    dB = mk(N * (ldk if b_kc else N) if b_kc else K * N)
    dC = mk(M * N)
    lda = ldk if a_kc else M
    ldb = ldk if b_kc else N
    ref = None
    for cfg in cfgs:
        lib.qemb_set_gemm_config(cfg); lib.qemb_set_gemm_ksplit(ks)
        run = lambda: check(lib.qemb_op_gemm(M, N, K, 1.0, dA.ptr, lda, a_kc, 0, dB.ptr, ldb, b_kc, 0, 0.0, dC.ptr, N, 0, 1))
        run(); run(); lib.qemb_sync()
        out = dC.numpy((M, N))[: min(M, 64), :256].copy()
        if ref is None:
            ref = out
        err = float(np.abs(out - ref).max())
        lib.qemb_timer_reset(5)
        for _ in range(reps):
            lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
        ms = C.c_double(); cnt = C.c_int64()
        lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
        t = ms.value / cnt.value * 1e-3
        fl = 2.0 * M * N * K
        print(json.dumps(dict(tag=tag, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, cfg=cfg, ksplit=ks, ms=round(t * 1e3, 4), tflops=round(fl / t / 1e12, 2),
                              frac_peak=round(fl / t / PEAK, 3), max_abs_diff_vs_first_cfg=err)), flush=True)
    lib.qemb_set_gemm_config(-1); lib.qemb_set_gemm_ksplit(0)
    for b in (dA, dB, dC):
        b.free()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    o, v = 20, 200
    npo, nmo, npv, nmv = o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2
    for bpc in (2,):
        t = C.c_double(); check(lib.qemb_mfma_f64_peak(40000, bpc, C.byref(t)))
        print(json.dumps(dict(tag="mfma_f64 register-only peak", blocks_per_cu=bpc, tflops=t.value)), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "clock":    # which of tile / shape / data sets the clock the chip holds? (with tools/gemm_pmc.sh)
        bench("(-) shape, 7x2 tile", nmo, nmv, nmv, 1, 1, (23,), ks=8, reps=reps)
        bench("(-) shape, 6x2 tile", nmo, nmv, nmv, 1, 1, (25,), ks=8, reps=reps)
        bench("(+) shape, 7x2 tile", npo, npv, npv, 1, 1, (23,), ks=8, reps=reps)
        bench("(+) shape M=190, 6x2 tile", 190, npv, npv, 1, 1, (25,), ks=8, reps=reps)
        bench("(+) shape M=224, 7x2 tile", 224, npv, npv, 1, 1, (23,), ks=8, reps=reps)
        bench("(-) shape, 6x2 tile again", nmo, nmv, nmv, 1, 1, (25,), ks=8, reps=reps)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":      # the subset profiled by tools/gemm_pmc.sh
        bench("pp-ladder (+) pairs", npo, npv, npv, 1, 1, (213, 23), ks=8, reps=reps)
        bench("pp-ladder (-) pairs", nmo, nmv, nmv, 1, 1, (215, 25), ks=8, reps=reps)
        bench("ph-ring (ov)^3 NT", o * v, o * v, o * v, 1, 1, (200, 4), reps=reps)
        sys.exit(0)
    bench("pp-ladder (+) pairs", npo, npv, npv, 1, 1, (213, 23), ks=8, reps=reps)
    bench("pp-ladder (-) pairs", nmo, nmv, nmv, 1, 1, (215, 25), ks=8, reps=reps)
    bench("tau-side dressing (+)", npo, o * v, npv, 1, 1, (213, 13), ks=8, reps=reps)
    bench("ph-ring (ov)^3 NN", o * v, o * v, o * v, 1, 0, (200, 0, 204, 4), reps=reps)
    bench("ph-ring (ov)^3 NT", o * v, o * v, o * v, 1, 1, (200, 0, 204, 4, 213, 13), reps=reps)
    bench("quarter transform C^T X^T", 220, 24310 * 220 // 4, 220, 0, 1, (213, 13), reps=reps)
    bench("slab . C batched-like", 220, 220 * 2000, 220, 1, 0, (213, 13), reps=reps)
    bench("U = t2 . Lvv", o * o * v, v, v, 1, 1, (200, 0, 1, 204, 4), reps=reps)
    bench("Woooo tau", o * o, v * v, o * o, 0, 0, (1, 200, 0), reps=reps)
