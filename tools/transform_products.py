"""The K = 220 products of the embedding -> MO transformation at n = 220 (four-index route) on their own, on their production tiles and the
alternatives that exist.  One JSON line per (product, cfg).  (Round 4's staggered-start instantiations were measured -- no gain,
profiles/r04_transform_products_stagger_*.jsonl -- and removed in round 5; q4f is round 5's flat form of the last quarter transform.)

    python tools/transform_products.py [reps]
"""
import ctypes as C
import json
import os
import sys

os.environ.setdefault("QEMB_GEMM_DIAGNOSTICS", "1")
import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = 220
nl = 224
npair = n * (n + 1) // 2
rng = np.random.default_rng(0)


def rand_dev(nelem):
    blk = rng.standard_normal(min(nelem, 1 << 22))
    d = DeviceBuffer(nelem)
    src = DeviceBuffer.from_numpy(blk)
    off = 0
    while off < nelem:
        m = min(blk.size, nelem - off)
        check(lib.qemb_d2d(d.at(off), src.ptr, m * 8))
        off += m
    src.free()
    return d


def timed(f):
    f(); f(); lib.qemb_sync()
    lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); f(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64()
    lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value


dC = DeviceBuffer.from_numpy(np.linalg.qr(rng.standard_normal((n, n)))[0])
X = rand_dev(npair * n * nl)
Y = rand_dev(npair * n * nl)
flop = 2.0 * n * n * npair * n
ref = {}
for tag, cfgs, call in (
    ("q1: Out[s',(pq r)] = C^T X, M=220 N=5.3M K=220 (A m-contig, B k-contig ld 224)", (13, 33),
     lambda: check(lib.qemb_op_gemm(n, npair * n, n, 1.0, dC.ptr, n, 0, 0, X.ptr, nl, 1, 0, 0.0, Y.ptr, npair * n, 0, 1))),
    ("q3: slab . C flat, M=5.3M N=220 K=220 (A k-contig ld 224, B n-contig)", (34,),
     lambda: check(lib.qemb_op_gemm(npair * n, n, n, 1.0, X.ptr, nl, 1, 0, dC.ptr, n, 0, 0, 0.0, Y.ptr, n, 0, 1))),
    ("q4: C^T . slab batched over 24310 pairs (A m-contig, B n-contig) -- rounds 1-4", (13, 33),
     lambda: check(lib.qemb_op_gemm(n, n, n, 1.0, dC.ptr, n, 0, 0, X.ptr, n, 0, n * n, 0.0, Y.ptr, n, n * n, npair))),
    ("q4f: the same slabs transposed, (slab^T) . C as ONE tall product over the rows (pair, q') through the slab-aware loader, M=5.3M N=220 K=220 -- round 5", (34,),
     lambda: check(lib.qemb_op_gemm_slab_rows(npair * n, n, n, X.ptr, n, n, n * n - n, dC.ptr, n, 0, Y.ptr, n, -1))),
):
    for cfg in cfgs:
        lib.qemb_set_gemm_config(cfg)
        ms = timed(call)
        out = np.empty(4096)
        check(lib.qemb_d2h(out.ctypes.data, Y.at(12345 * 8), 4096 * 8))
        key = tag.split(':')[0]
        same = None
        if out is not None:
            if key not in ref:
                ref[key] = out
            same = bool(np.array_equal(out, ref[key]))
        print(json.dumps(dict(product=tag, cfg=cfg, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2), frac_of_78_6=round(flop / ms / 1e9 / 78.6, 3),
                              same_as_first_cfg=same)), flush=True)
    lib.qemb_set_gemm_config(-1)
