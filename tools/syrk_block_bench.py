"""The block products of df_pair_product (ao2mo.cpp) on their own: S[P1][P2] = sum_L B[L,P1] B[L,P2] for one block column of the n = 220 factor route
(M = npair - c0, N = 3072, K = naux), with the factor stored [naux][npair] (operands M/N-contiguous) and [npair][naux] (K-contiguous), per tile config.

    python tools/syrk_block_bench.py [naux]
"""
import sys

sys.path.insert(0, "tools")
from gemm_bench import bench

naux = int(sys.argv[1]) if len(sys.argv) > 1 else 660
np_ = 24310
for M in (np_, np_ - 4 * 3072):
    for kc in (0, 1):
        for cfg in (-1, 0, 4, 13):
            bench(M, 3072, naux, kc, kc, cfg, tag="pair product block, factor %s" % ("[npair][naux]" if kc else "[naux][npair]"))
