"""Rehearse bench.py's host logic without a GPU: the product library handle is replaced by the scalar mock of tests/hostcheck
(test infrastructure) and the workload shrunk.  A development aid for the GPU-less build container -- never a measurement.

    python tools/bench_rehearsal.py [bench.py arguments, e.g. --n 26 --nocc 5 --frags-per-gpu 3 --nstreams 2]
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests" / "hostcheck"):
    sys.path.insert(0, str(p))
import build as hc_build  # noqa: E402
from quemb_amd import _lib  # noqa: E402

_lib._lib = _lib.declare(C.CDLL(str(hc_build.build())))
import bench  # noqa: E402

if __name__ == "__main__":
    if "--n" not in sys.argv:       # the scalar mock is only usable on a shrunken workload
        sys.argv[1:1] = ["--n", "26", "--nocc", "5", "--frags-per-gpu", "3", "--nstreams", "2", "--steps", "1", "--warmup", "1", "--roofline-iters", "2"]
    bench.main()
