"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters on THIS code's access patterns (MI355X guide: 'calibrate on a known byte
count in your own access pattern'): streaming kernels with known traffic and known access width --
    lincomb_kernel      8-byte loads / stores per lane (512 B per wave instruction): out = a x + b y over 2^27 doubles (reads 2 GiB, writes 1 GiB)
    diis_push<true>     16-byte loads / stores (tools/hbm_pmc.py already shows it at 0.96 x algorithmic with the x2 correction)
Run under the two --pmc passes; tools/pmc_calibrate.py --report <fetch dir> <write dir> prints bytes-per-counter-unit factors."""
import csv, glob, json, sys

if "--report" in sys.argv:
    i = sys.argv.index("--report")
    out = {}
    for d, counter in ((sys.argv[i + 1], "FETCH_SIZE"), (sys.argv[i + 2], "WRITE_SIZE")):
        f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "lincomb_kernel" in r["Kernel_Name"]]
        big = [v for v in vals if v > 0.5 * max(vals)]
        out[counter + "_KB_mean"] = sum(big) / len(big); out[counter + "_launches"] = len(big)
    n = 1 << 27
    out["known_read_bytes"] = 2.0 * n * 8; out["known_write_bytes"] = 1.0 * n * 8
    out["read_bytes_per_FETCH_SIZE_byte"] = out["known_read_bytes"] / (out["FETCH_SIZE_KB_mean"] * 1024.0)
    out["write_bytes_per_WRITE_SIZE_byte"] = out["known_write_bytes"] / (out["WRITE_SIZE_KB_mean"] * 1024.0)
    out["note"] = "8-byte-per-lane coalesced loads / stores (lincomb_kernel, 2^27 doubles per vector): how many real bytes one counted byte stands for"
    print(json.dumps(out, indent=1))
    sys.exit(0)

sys.path.insert(0, ".")
import numpy as np
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check
lib = _lib.init(0)
n = 1 << 27
x, y, o = DeviceBuffer(n), DeviceBuffer(n), DeviceBuffer(n)
src = DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal(1 << 22))
for b in (x, y):
    for off in range(0, n, 1 << 22):
        check(lib.qemb_d2d(b.at(off), src.ptr, (1 << 22) * 8))
for _ in range(6):
    check(lib.qemb_op_lincomb2(n, 0.5, x.ptr, 0.25, y.ptr, 0.0, o.ptr))
lib.qemb_sync()
print("calibration workload done")
