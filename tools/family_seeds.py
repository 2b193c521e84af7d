"""Which fragments of the synthetic family (bench.py: seed = 20260803 + I, n = 220, n_occ = 20, ERI scale 0.03) does the device solve?  HOMO-LUMO gap of the
fragment RHF, CCSD iterations, or the failure -- for I = 0 .. 63 (BASELINE configs[2]: 64 fragments over 8 GPUs)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench
from quemb_amd import _lib
from quemb_amd._lib import QembError
from quemb_amd.fragsolver import DeviceFragment, default_opts

lib = _lib.init(0)
n, o = 220, 20
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 64)
opts = default_opts(lib)
for I in range(lo, hi):
    h, veff0, d4, dB, naux = bench.make_device_eris(lib, n, bench.SEED0 + I, 0.03, block=False)
    fr = DeviceFragment(n, 22, lib=lib)
    fr.set_df_only_dev(dB.ptr, naux); dB.free()
    fr.set_energy_data(h, veff0, None, 1.0, [6, 7])
    t0 = time.time()
    row = dict(I=I, seed=bench.SEED0 + I)
    try:
        r = fr.scf(o, h, None, opts=opts)
        row.update(gap=float(r["mo_energy"][o] - r["mo_energy"][o - 1]), scf_cycles=r["cycles"])
        dm0 = 2.0 * r["mo_coeff"][:, :o] @ r["mo_coeff"][:, :o].T
        out = fr.solve(o, h, dm0, opts=opts, eeval=True)
        row.update(n_iter=out["n_iter"], e_corr=out["e_corr_mo"])
    except QembError as e:
        row.update(failed=str(e)[-80:])
    row["s"] = round(time.time() - t0, 2)
    print(json.dumps(row), flush=True)
    fr.free()
