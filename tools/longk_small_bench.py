"""Long-K products with a small result inside the n = 220 CCSD iteration, per tile configuration and K split (the split's reduction pass is inside the figure:
the application leaves the slabs to the consumer instead):  Xw (+): M = 210, N = 400, K = 20100 (both operands K-contiguous);  Fvv': M = N = 200, K = 80000 (both
operands stored [K][M] / [K][N]).

    python tools/longk_small_bench.py
"""
import sys

sys.path.insert(0, "tools")
from gemm_bench import bench, lib

for (M, N, K, kc, tag, combos) in (
        (210, 400, 20100, 1, "Xw(+)", ((1, 19), (1, 32), (0, 32), (0, 64), (13, 32), (13, 64), (33, 64), (4, 64), (4, 128))),
        (200, 200, 80000, 0, "Fvv'", ((1, 32), (1, 48), (1, 64), (0, 64), (0, 128), (4, 256), (13, 128), (13, 256)))):
    for cfg, ks in combos:
        lib.qemb_set_gemm_ksplit(ks)
        bench(M, N, K, kc, kc, cfg, tag="%s ksplit=%d" % (tag, ks))
lib.qemb_set_gemm_ksplit(0)
