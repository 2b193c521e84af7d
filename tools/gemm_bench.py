"""GPU micro-benchmark of the FP64 MFMA GEMM on the shapes of the hot path (run via gpurun)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
PEAK = 78.6e12


def bench(M, N, K, a_kc, b_kc, cfg, reps=5, tag=""):
    rng = np.random.default_rng(0)
    # fill on host in chunks (random data: guide rule 25)
    def mk(n):
        b = DeviceBuffer(n)
        chunk = rng.standard_normal(min(n, 1 << 24))
        off = 0
        while off < n:
            m = min(chunk.size, n - off)
            check(lib.qemb_h2d(b.at(off), chunk.ctypes.data, m * 8))
            off += m
        return b
    dA, dB, dC = mk(M * K), mk(K * N), mk(M * N)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    lib.qemb_set_gemm_config(cfg)
    def run():
        check(lib.qemb_op_gemm(M, N, K, 1.0, dA.ptr, lda, a_kc, 0, dB.ptr, ldb, b_kc, 0, 0.0, dC.ptr, N, 0, 1))
    run(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); run(); lib.qemb_timer_end(5)
    import ctypes as C
    ms = C.c_double(); cnt = C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    t = ms.value / cnt.value * 1e-3
    fl = 2.0 * M * N * K
    out = dict(tag=tag, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, cfg=cfg, ms=t * 1e3, tflops=fl / t / 1e12, frac_peak=fl / t / PEAK)
    print(json.dumps(out), flush=True)
    lib.qemb_set_gemm_config(-1)
    for b in (dA, dB, dC): b.free()
    return out


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    v = 120 if quick else 200
    o = 20
    import ctypes as C
    for bpc in (1, 2, 4):
        t = C.c_double(); check(lib.qemb_mfma_f64_peak(40000, bpc, C.byref(t)))
        print(json.dumps(dict(tag="mfma_f64 register-only peak", blocks_per_cu=bpc, tflops=t.value)), flush=True)
    npair = o * (o + 1) // 2
    if len(sys.argv) > 1 and sys.argv[1] == "ladder":
        npv, nmv = v * (v + 1) // 2, v * (v - 1) // 2
        for ks in (8,):
            lib.qemb_set_gemm_ksplit(ks)
            for cfg in (10, 13):
                r1 = bench(npair, npv, npv, 1, 1, cfg, tag="pp-ladder (+) block ksplit=%d" % ks)
            for cfg in (10, 13, 15):
                r2 = bench(o * (o - 1) // 2, nmv, nmv, 1, 1, cfg, tag="pp-ladder (-) block ksplit=%d" % ks)
        lib.qemb_set_gemm_ksplit(0)
        sys.exit(0)
    bench(o * o, v * v, v * v, 1, 1, 1, tag="pp-ladder dense tau[ij,cd] W[ab,cd]")
    npv, nmv = v * (v + 1) // 2, v * (v - 1) // 2
    lib.qemb_set_gemm_ksplit(8)
    r1 = bench(npair, npv, npv, 1, 1, 10, tag="pp-ladder (+) block M=npair(o) N=K=npair(v), ksplit=8")
    r2 = bench(o * (o - 1) // 2, nmv, nmv, 1, 1, 10, tag="pp-ladder (-) block, ksplit=8")
    print(json.dumps(dict(tag="(+/-) ladder total", ms=r1["ms"] + r2["ms"], dense_equivalent_tflops=2.0 * o * o * v ** 4 / ((r1["ms"] + r2["ms"]) * 1e-3) / 1e12)), flush=True)
    lib.qemb_set_gemm_ksplit(0)
    bench(o, v, o * v * v, 1, 0, -1, tag="t1 term: M=o N=v K=o v^2 (split-K)")
    bench(v, v, o * o * v, 0, 0, -1, tag="Fvv: M=v N=v K=o^2 v (split-K)")
    bench(o * o, o * o, v * v, 1, 1, -1, tag="Woooo: M=N=o^2 K=v^2 (split-K)")
    for cfg in (0, 4):
        bench(o * v, o * v, o * v, 1, 0, cfg, tag="ph-ring (ov)^3")
        bench(4096, 4096, 4096, 1, 1, cfg, tag="square NT")
        bench(4096, 4096, 4096, 1, 0, cfg, tag="square NN")
        bench(4096, 4096, 4096, 0, 0, cfg, tag="square TN")
    for cfg in (0, 4, 10):
        bench(220, 220 ** 3, 220, 0, 1, cfg, tag="quarter transform C^T X^T")
