"""Fragment-size sweep of the synthetic family: for each (n, n_occ) the CCSD iteration rate and the executed TFLOP/s of a whole
iteration, one fragment on one stream and several fragments in the best mode (lock step for small fragments, streams otherwise).

    python tools/size_sweep.py                  # the sizes bench.py reports as `size_sweep`
    python tools/size_sweep.py 96:9 132:12      # chosen sizes

`run(lib, sizes)` is what bench.py calls.  Executed flops of an iteration = 2 M N K batch summed over every FP64 MFMA product one eager update_amps
issues (qemb_gemm_flop_count): the (+/-) pair-packed ladder counts with its quarter of the dense flops, padding of tiles does not count.
"""
import ctypes as C
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
PEAK = 78.6
SIZES = ((42, 21), (64, 6), (96, 9), (132, 12), (176, 16), (220, 20), (300, 30))


def _fragment(lib, n, seed):
    from quemb_amd._lib import DeviceBuffer, check
    from quemb_amd.fragsolver import DeviceFragment
    rng = np.random.default_rng(seed)
    naux = 3 * n
    scale = 0.06 * min(1.0, (55.0 / n) ** 0.5)
    B = scale * rng.standard_normal((naux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    npair = Bp.shape[1]
    dB = DeviceBuffer.from_numpy(Bp, lib=lib)
    d4 = DeviceBuffer(npair * npair, lib=lib)
    check(lib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1), lib=lib)
    A = rng.standard_normal((n, n))
    h = np.diag(2.0 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    nf = max(2, min(22, n // 2))
    fr = DeviceFragment(n, nf, lib=lib)
    fr.set_eri_s4_dev(d4.ptr); d4.free()
    fr.set_df_factor_dev(dB.ptr, naux); dB.free()
    V = np.random.default_rng(1).standard_normal((n, n))
    fr.set_energy_data(h, 0.05 * (V + V.T), None, 1.0, list(range(min(4, nf))))
    return fr, h


def _timer(lib, slot, nctx=1, reset=0):
    ms_t, cnt_t = 0.0, 0
    for k in range(nctx):
        ms, cnt = C.c_double(), C.c_int64()
        if lib.qemb_ctx_timer_read(k, slot, C.byref(ms), C.byref(cnt), reset) == 0:
            ms_t += ms.value; cnt_t += cnt.value
    return ms_t, cnt_t


def one_size(lib, n, o, nbest=None, lockstep_upto=None):
    from quemb_amd.fragsolver import default_opts, solve_batch
    from quemb_amd._lib import QembError, check
    import os
    v = n - o
    if lockstep_upto is None:
        lockstep_upto = int(os.environ.get("QEMB_SWEEP_LOCKSTEP_UPTO", "64"))
    if nbest is None:
        nbest = int(os.environ.get("QEMB_SWEEP_NBEST", "0")) or (6 if n <= 64 else (4 if n <= 256 else 2))
    opts = default_opts(lib)
    # fragments of the family whose RHF / CCSD converge from the core guess (a few seeds of the mid sizes do not at this ERI strength: skipped, and said so)
    frs, dm0s, skipped, seed = [], [], [], 20260803
    while len(frs) < nbest and seed < 20260803 + 4 * nbest + 4:
        f, hh = _fragment(lib, n, seed)
        try:
            oo = f.solve(o, hh, opts=opts, eeval=True)
            Cq = oo["mo_coeff"]; dm0s.append(2.0 * Cq[:, :o] @ Cq[:, :o].T); frs.append((f, hh))
        except QembError:
            skipped.append(seed); f.free()
        seed += 1
    if not frs:
        raise RuntimeError(f"no fragment of n={n} converged")
    nbest = len(frs)
    fr, h = frs[0]
    # executed flops of one update: prepare, then ONE eager iteration with the counter around it
    fr.prepare_ccsd(o, h, opts=opts)
    fl = C.c_double()
    lib.qemb_gemm_flop_count(C.byref(fl), 1)
    fr.ccsd_iterate(1)
    lib.qemb_gemm_flop_count(C.byref(fl), 1)
    flop_iter = fl.value
    fr.ccsd_reset()
    # single stream: the timed solve the way a sweep meets the fragment (previous density, potential moved a little)
    dm0 = dm0s[0]
    h2 = h.copy(); h2[:2, :2] += 1e-3
    fr.solve(o, h2, dm0=dm0, opts=opts, eeval=True)            # (graph / pool warm)
    fr.solve(o, h2, dm0=dm0, opts=opts, eeval=True)
    lib.qemb_device_sync(); _timer(lib, 2, 1, reset=1)
    reps = 3 if n <= 132 else 1
    t0 = time.perf_counter(); nit = 0
    for _ in range(reps):
        o1 = fr.solve(o, h2, dm0=dm0, opts=opts, eeval=True); nit += o1["n_iter"]
    lib.qemb_device_sync(); wall = time.perf_counter() - t0
    it_ms, it_cnt = _timer(lib, 2, 1)
    iter_ms = it_ms / max(it_cnt, 1)
    single = dict(iterations_per_s=nit / wall, solve_ms=wall / reps * 1e3, iteration_ms=iter_ms, iterations=nit // reps,
                  tflops_iteration=flop_iter / (iter_ms * 1e-3) / 1e12 if iter_ms > 0 else None)
    single["frac_of_peak"] = single["tflops_iteration"] / PEAK if single["tflops_iteration"] else None
    # best mode: nbest fragments at once
    hs = [x[1] for x in frs]
    lockstep = n <= lockstep_upto and nbest > 1
    if lockstep:
        run = lambda: solve_batch([x[0] for x in frs], [o] * nbest, hs, dm0s, opts=opts, eeval=True)
    else:
        have = lib.qemb_ctx_count(nbest + 1)
        nw = max(1, min(nbest, have - 1))

        def work(w):
            if nw > 1:
                check(lib.qemb_ctx_bind(w + 1), "qemb_ctx_bind", lib)
            return [frs[k][0].solve(o, hs[k], dm0=dm0s[k], opts=opts, eeval=True) for k in range(w, nbest, nw)]

        def run():
            with ThreadPoolExecutor(max_workers=nw) as pool:
                return [r for part in pool.map(work, range(nw)) for r in part]
    run(); run()
    lib.qemb_device_sync()
    t0 = time.perf_counter(); nitb = 0
    for _ in range(reps):
        nitb += sum(r["n_iter"] for r in run())
    lib.qemb_device_sync(); wallb = time.perf_counter() - t0
    best = dict(mode=("lockstep" if lockstep else "streams") + f" x{nbest}", iterations_per_s=nitb / wallb, sweep_ms=wallb / reps * 1e3,
                tflops_whole_solve=nitb * flop_iter / wallb / 1e12)
    best["frac_of_peak"] = best["tflops_whole_solve"] / PEAK
    single["tflops_whole_solve"] = nit * flop_iter / wall / 1e12
    if single["tflops_whole_solve"] > best["tflops_whole_solve"]:      # (n = 300: two fragments in flight contend for the chip and lose to one)
        best = dict(mode="single stream (several in flight were slower: " + best["mode"] + f" {best['iterations_per_s']:.1f} it/s)", iterations_per_s=single["iterations_per_s"],
                    sweep_ms=single["solve_ms"], tflops_whole_solve=single["tflops_whole_solve"], frac_of_peak=single["tflops_whole_solve"] / PEAK)
    for f, _ in frs:
        f.free()
    lib.qemb_trim_all()      # the next size has other block sizes: nothing parked here is reused
    row = dict(n=n, n_occ=o, n_virt=v, executed_flop_per_iteration=flop_iter, single_stream=single, best_mode=best)
    if skipped:
        row["seeds_skipped_not_converging"] = skipped
    return row


def run(lib, sizes=SIZES, log=None):
    rows = []
    for n, o in sizes:
        t0 = time.perf_counter()
        try:
            rows.append(one_size(lib, n, o))
        except Exception as e:  # noqa: BLE001
            rows.append(dict(n=n, n_occ=o, failed=str(e)))
        if log:
            log(f"size sweep n={n}: {time.perf_counter() - t0:.1f} s")
    return dict(peak_tflops=PEAK, rows=rows,
                what="synthetic family of SURVEY 8d per fragment size: single_stream = one fragment, one stream (iteration_ms = HIP-event bracket of a whole CCSD iteration, "
                     "tflops_iteration = executed FP64 product flops of one amplitude update / iteration_ms; iterations_per_s over whole warm solves incl. fragment RHF, MO integrals, "
                     "RDMs and energies); best_mode = several fragments of that size at once (lock step up to n = 64, separate streams beyond -- eight n = 96 / 132 fragments in lock step were measured no better than four streams; the single-stream figure when that is the better one), "
                     "tflops_whole_solve = iterations x executed flop per iteration / wall time of the whole solves")


if __name__ == "__main__":
    from quemb_amd import _lib
    lib = _lib.init(0)
    sizes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if ":" in a] or SIZES
    res = run(lib, sizes, log=lambda m: print(m, file=sys.stderr, flush=True))
    for r in res["rows"]:
        print(json.dumps(r), flush=True)
