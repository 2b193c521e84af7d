"""Where do the waves of the 8-wave GEMM tiles spend their cycles?  Diagnostic instantiations (tile configs 313 / 315 / 304) stamp
s_memtime around the per-tile barrier; per tile of BK = 16 a wave issues (WM x WN) x 4 MFMAs of 64 cycles, two waves share a SIMD.
    python tools/gemm_stamps.py
"""
import ctypes as C
import os
os.environ.setdefault("QEMB_GEMM_DIAGNOSTICS", "1")   # this tool runs the stamping instantiations (tile configs 3xx)
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
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


def run(tag, M, N, K, cfg, ks, wm, wn):
    dA, dB, dC = mk(M * K), mk(N * K), mk(M * N)
    out = (C.c_double * 11)()
    for _ in range(3):
        check(lib.qemb_op_gemm_stamps(M, N, K, dA.ptr, K, dB.ptr, K, dC.ptr, N, cfg, ks, out), "qemb_op_gemm_stamps", lib)
    kchunk = (K + max(ks, 1) - 1) // max(ks, 1)
    ntile = (kchunk + 15) // 16
    work, bar, last = out[0] / ntile, out[1] / ntile, out[2] / ntile
    ideal = wm * wn * 4 * 64 * 2          # MFMA cycles of both waves of a SIMD per tile
    print(json.dumps(dict(tag=tag, cfg=cfg, ms=round(out[3], 4), tflops=round(2.0 * M * N * K / out[3] / 1e9, 2), k_tiles_per_wave=ntile,
                          cycles_per_tile=dict(ksteps_0_to_2=round(work, 1), barrier_segment=round(bar, 1), last_kstep=round(last, 1),
                                               total=round(work + bar + last, 1), mfma_pipe_ideal=ideal),
                          barrier_share=round(bar / (work + bar + last), 4), ideal_share=round(ideal / (work + bar + last), 4),
                          per_wave_cycles=dict(main_loop=round(out[0] + out[1] + out[2]), entry_to_loop_end=round(out[4]), entry_to_exit=round(out[5])),
                          prologue_cycles=dict(entry_to_first_load=round(out[7]), tile0_load_and_store=round(out[8]), fetch_tile1=round(out[9]), barrier=round(out[10])),
                          workgroups=int(out[6]), implied_clock_ghz=round(out[5] * (out[6] / 256.0) / (out[3] * 1e6), 3))), flush=True)
    for b in (dA, dB, dC):
        b.free()


if __name__ == "__main__":
    o, v = 20, 200
    npo, nmo, npv, nmv = o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2
    run("pp-ladder (+) pairs", npo, npv, npv, 313, 8, 7, 2)
    run("pp-ladder (-) pairs", nmo, nmv, nmv, 315, 8, 6, 2)
    run("ph-ring (ov)^3 NT", o * v, o * v, o * v, 304, 0, 4, 4)
    # round 4: the short-K products of the MO transformation (K = n = 220: 14 k-tiles per 224 x 128 output tile), a quarter of the columns
    n = 220
    run("MO quarter transform, K = 220 (M = 220, N = 128 x 10240)", n, 128 * 10240, n, 313, 0, 7, 2)
