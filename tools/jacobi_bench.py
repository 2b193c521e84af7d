"""GPU: symmetric eigensolver (wavefront Jacobi) accuracy and time at the sizes of a fragment Fock matrix, cold and from a nearly
diagonal start.  (profiles/r02_jacobi_fused_experiment.jsonl was taken with an experiment build that also had an all-rounds-in-one-launch
kernel, selected unless QEMB_JACOBI_FUSED=0; the shipped library has the one-launch-per-round path only and ignores the variable.)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check

lib = _lib.init(0)
rng = np.random.default_rng(1)
sizes = [int(a) for a in sys.argv[1:]] or [97, 130, 220, 300, 511, 512, 600]
for n in sizes:
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T) + np.diag(np.arange(n) * 0.7)
    w_ref = np.linalg.eigvalsh(A)
    out = {}
    for label, M in (("cold", A), ("warm", np.diag(w_ref) + 1e-4 * 0.5 * (lambda X: X + X.T)(rng.standard_normal((n, n))))):
        dA, dw, dV = DeviceBuffer.from_numpy(M), DeviceBuffer(n), DeviceBuffer(n * n)
        sw = C.c_int()
        check(lib.qemb_op_jacobi_eigh(n, dA.ptr, dw.ptr, dV.ptr, C.byref(sw)))        # warm-up (allocations)
        check(lib.qemb_h2d(dA.ptr, np.ascontiguousarray(M).ctypes.data, n * n * 8))
        lib.qemb_sync(); t0 = time.perf_counter()
        check(lib.qemb_op_jacobi_eigh(n, dA.ptr, dw.ptr, dV.ptr, C.byref(sw)))
        lib.qemb_sync(); dt = time.perf_counter() - t0
        w, V = dw.numpy((n,)), dV.numpy((n, n))
        res = np.abs(M @ V - V * w).max(); orth = np.abs(V.T @ V - np.eye(n)).max()
        out[label] = dict(ms=round(dt * 1e3, 3), sweeps=sw.value, residual=float(res), orth=float(orth), eig_err=float(np.abs(np.sort(w) - np.linalg.eigvalsh(M)).max()))
        for b in (dA, dw, dV):
            b.free()
    print(json.dumps(dict(n=n, block_rounds=os.environ.get("QEMB_JACOBI_BLOCK", "1"), **out)), flush=True)
