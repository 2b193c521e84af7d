import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check, i64x4
lib = _lib.init(0)
rng = np.random.default_rng(0)
def timed(f, reps=20):
    f(); f(); lib.qemb_sync(); lib.qemb_timer_reset(5)
    for _ in range(reps):
        lib.qemb_timer_begin(5); f(); lib.qemb_timer_end(5)
    ms = C.c_double(); cnt = C.c_int64(); lib.qemb_timer_read(5, C.byref(ms), C.byref(cnt)); return ms.value / cnt.value
o = 20
for v in (192, 200, 208, 224):
    N2 = o * o * v * v
    t2 = DeviceBuffer.from_numpy(rng.standard_normal(N2)); t1 = DeviceBuffer.from_numpy(rng.standard_normal(o * v))
    outs = [DeviceBuffer(N2) for _ in range(6)]
    t = timed(lambda: check(lib.qemb_op_ccsd_ph_layouts(o, v, t2.ptr, t1.ptr, *[b.ptr for b in outs])))
    print(json.dumps(dict(kernel="ccsd_ph_layouts", v=v, ms=round(t, 4), TBps=round(7 * N2 * 8 / t / 1e9, 2))), flush=True)
    # perm4 [i,j,a,b] -> [i,a,j,b] (copy4 transpose class) with accumulate: out += in
    dims = i64x4((o, v, o, v)); si = i64x4((o * v * v, v, v * v, 1)); so = i64x4((v * o * v, o * v, v, 1))
    t = timed(lambda: check(lib.qemb_op_copy4(dims, t2.ptr, si, outs[0].ptr, so, 1.0, 1.0)))
    print(json.dumps(dict(kernel="copy4 [i,j,a,b]->[i,a,j,b] accumulate", v=v, ms=round(t, 4), TBps=round(3 * N2 * 8 / t / 1e9, 2))), flush=True)
    # transposing perm: out[i,j,a,b] += in[i,b,j,a]
    dims = i64x4((o, o, v, v)); si = i64x4((v * o * v, v, 1, o * v)); so = i64x4((o * v * v, v * v, v, 1))
    t = timed(lambda: check(lib.qemb_op_copy4(dims, t2.ptr, si, outs[1].ptr, so, -1.0, 1.0)))
    print(json.dumps(dict(kernel="copy4 out[i,j,a,b] -= in[i,b,j,a] (transpose)", v=v, ms=round(t, 4), TBps=round(3 * N2 * 8 / t / 1e9, 2))), flush=True)
    for b in outs + [t2, t1]: b.free()
