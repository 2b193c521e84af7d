"""GPU experiment: what does the clock the chip holds under the MODE 1 GEMM main loop owe to its memory traffic?

Ablation instantiations of the ladder (224 x 128) and ring (128 x 256) tiles -- WRONG results by construction -- are timed on the shapes
of one CCSD iteration: 3 = no global loads in the main loop, 4 = every other A fragment is not re-read from LDS, 5 = both.  The MFMA
count and the barrier structure are unchanged, so a shorter run time is a higher clock (or fewer stalls) bought by the removed traffic:
the ceiling on what a larger wave tile (fewer LDS bytes per MFMA) or more L2 reuse could give.  Second part: the K = 220 products of
the MO transformation on the 224 x 128 tile (one 8-wave workgroup per CU) against the 112 x 128 tile (two 4-wave workgroups per CU).

    python tools/gemm_ablation.py [reps]
"""
import os
import sys
os.environ.setdefault("QEMB_GEMM_DIAGNOSTICS", "1")   # this tool runs the ablation instantiations (tile configs 4xx-6xx)

sys.path.insert(0, ".")
from tools.gemm_modes import bench  # noqa: E402  (initialises the device)

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    o, v = 20, 200
    npo, npv, nov = o * (o + 1) // 2, v * (v + 1) // 2, o * v
    bench("pp-ladder (+): full / no global loads / half A reads / both", npo, npv, npv, 1, 1, (23, 413, 513, 613, 23), ks=8, reps=reps)
    bench("ph-ring (ov)^3: full / no global loads / half A reads / both", nov, nov, nov, 1, 1, (4, 404, 504, 604, 4), reps=reps)
    n, npair = 220, 220 * 221 // 2
    bench("quarter transform C^T X^T (224x128 vs 2 x 112x128)", n, npair * n // 4, n, 0, 1, (13, 33, 13, 33), reps=reps)
    bench("quarter transform NN", n, npair * n // 4, n, 0, 0, (13, 33), reps=reps)
    bench("quarter transform TN", n, npair * n // 4, n, 1, 0, (13, 33), reps=reps)
    bench("slab . C flat (tall, N = 220): 224x128 / 128x224 / 128x256 / 112x128", npair * n // 4, n, n, 1, 0, (13, 34, 4, 33), reps=reps)
