#!/usr/bin/env python
"""bench.py -- fragment-sweep throughput of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the one the metric's n_occ/n_virt and the 1/2/4/8-GPU scaling are quoted on;
configs[1] -- octane BE2 -- is a parity case in tests/): F synthetic fragments PER GPU (weak scaling), each
n = 220 embedding orbitals, n_occ = 20, n_virt = 200, DF-factorised 8-fold-symmetric ERIs (SURVEY.md 8d family,
seed 20260803 + global fragment index; ERI scale 0.03), ERIs resident in HBM before timing.

One STEP = one be_func sweep (one objective evaluation of the density-matching loop, molbe/solver.py:244) over the
rank's fragments: per fragment  fragment RHF -> embedding->MO integral transform -> RCCSD to convergence ->
1-RDM -> fragment energy;  then ONE all-reduce (RCCL) of the residual/energy buffer.  Nothing is cached between
steps (amplitudes restart from MP2 exactly like the reference).
value = CCSD iterations completed by all ranks in the K timed steps / wall time (max over ranks).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the host driver of this pool only supports dmabuf IPC: RCCL needs this before the runtime starts (already exported on the boxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X FP64 matrix peak (AMD spec; == the FP64 vector peak). The MI355X guide
                                 # lists no f64 row; see DESIGN.md "Roofline".
SEED0 = 20260803


T_START = time.perf_counter()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frags-per-gpu", type=int, default=8)
    ap.add_argument("--nstreams", type=int, default=1, help="fragments in flight per GPU (separate HIP streams); the default 1 keeps the "
                    "ladder kernel's HIP-event / rocprofv3 durations uncontended")
    ap.add_argument("--n", type=int, default=220)
    ap.add_argument("--nocc", type=int, default=20)
    ap.add_argument("--scale", type=float, default=0.03)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3, help="amplitude updates timed by the CPU baseline (about 4 s each on 16 threads)")
    ap.add_argument("--cpu-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=16, help="BLAS threads of the CPU baseline (the box's CPU share of one GPU)")
    return ap.parse_args()


def make_fragment(lib, n, nf, seed, scale):
    """Synthetic fragment: h on the host, ERIs built on the device from the DF factor and left resident."""
    from quemb_amd._lib import DeviceBuffer, check
    from quemb_amd.fragsolver import DeviceFragment
    rng = np.random.default_rng(seed)
    naux = 3 * n
    B = scale * rng.standard_normal((naux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    il = np.tril_indices(n)
    Bp = np.ascontiguousarray(B[:, il[0], il[1]])
    npair = Bp.shape[1]
    dB = DeviceBuffer.from_numpy(Bp)
    d4 = DeviceBuffer(npair * npair)
    check(lib.qemb_op_gemm(npair, npair, naux, 1.0, dB.ptr, npair, 0, 0, dB.ptr, npair, 0, 0, 0.0, d4.ptr, npair, 0, 1))
    A = rng.standard_normal((n, n))
    h = np.diag(2.0 * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    fr = DeviceFragment(n, nf)
    fr.set_eri_s4_dev(d4.ptr)
    dB.free(); d4.free()
    V = rng.standard_normal((n, n)); veff0 = 0.05 * (V + V.T)
    fr.set_energy_data(h, veff0, None, 1.0, list(range(nf)))
    return fr, h, B


def read_timer(lib, slot, nctx=1):
    """device timer `slot` summed over the execution contexts 0..nctx-1"""
    tot, n = 0.0, 0
    for k in range(nctx):
        ms = C.c_double(); cnt = C.c_int64()
        lib.qemb_ctx_timer_read(k, slot, C.byref(ms), C.byref(cnt), 0)
        tot += ms.value; n += cnt.value
    return tot, n


def cpu_baseline(fr, h, dm0, o, opts, iters, threads, timeout_s=300):
    """The oracle ('port') on the host cores, timed on a bounded sample of the SAME workload: fragment 0 of this rank,
    its MO integrals exported from the device (so no CPU time goes into re-deriving inputs), `iters` full RCCSD
    amplitude updates (oracle/qemb_oracle/ccsd_lean.py, NumPy/BLAS) starting from the MP2 guess.  Runs in a child
    process with the BLAS thread count pinned through the environment and a hard timeout."""
    import shutil
    import subprocess
    import tempfile
    n = fr.n
    v = n - o
    fr.prepare_ccsd(o, h, dm0, opts=opts)
    shapes = dict(oooo=(o, o, o, o), ovoo=(o, v, o, o), ovov=(o, v, o, v), ovvv=(o, v, v, v), Vl=(v, v, v, v),
                  W1base=(o, v, o, v), W2base=(o, v, o, v), eo=(o,), ev=(v,))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="qemb_bench_", dir=base)
    try:
        for name, shp in shapes.items():
            np.save(os.path.join(d, name + ".npy"), fr.ccsd_export(name, shp))
        log(f"cpu_baseline: integrals exported to {d}; running {iters} amplitude update(s) on {threads} threads")
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-worker", d, "--cpu-iters", str(iters), "--nocc", str(o)],
                           env=env, capture_output=True, text=True, timeout=timeout_s)
        if p.returncode != 0:
            return dict(value=None, unit="CCSD iterations/s", cores=threads, kind="port", sample=f"worker failed: {p.stderr[-400:]}")
        r = json.loads(p.stdout.strip().splitlines()[-1])
        return dict(value=1.0 / r["s_per_iteration"], unit="CCSD iterations/s", cores=threads, kind="port",
                    sample=f"fragment 0 of the timed workload (n_occ={o}, n_virt={v}); MO integrals exported from the device; "
                           f"{iters} full RCCSD amplitude update(s) from the MP2 guess by oracle/qemb_oracle/ccsd_lean.py (NumPy/BLAS, {threads} threads)",
                    s_per_iteration=r["s_per_iteration"], e_corr_after_sample=r["e_corr"])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="CCSD iterations/s", cores=threads, kind="port", sample=f"worker exceeded {timeout_s} s")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_worker(d, o, iters):
    """child process of cpu_baseline: load the exported blocks, time the oracle's amplitude update."""
    sys.path.insert(0, str(ROOT / "oracle"))
    from qemb_oracle import ccsd_lean
    ld = lambda nm: np.load(os.path.join(d, nm + ".npy"))
    eo, ev = ld("eo"), ld("ev")
    W1, W2 = ld("W1base"), ld("W2base")
    ovvo = np.ascontiguousarray(W1.transpose(2, 3, 1, 0))     # ovvo[k,c,a,i] = W1base[i,a,k,c]
    oovv = np.ascontiguousarray(W2.transpose(2, 0, 1, 3))     # oovv[k,i,a,c] = W2base[i,a,k,c]
    er = ccsd_lean.LeanEris.from_blocks(o, np.concatenate([eo, ev]), ld("oooo"), ld("ovoo"), ld("ovov"), oovv, ovvo, ld("ovvv"), ld("Vl"))
    eia = eo[:, None] - ev[None, :]
    t1 = np.zeros((o, len(ev))); t2 = er.ovov.transpose(0, 2, 1, 3) / (eia[:, None, :, None] + eia[None, :, None, :])
    t0 = time.perf_counter()
    for _ in range(iters):
        t1, t2 = ccsd_lean.update_amps(t1, t2, er)
    dt = (time.perf_counter() - t0) / iters
    tau = t2 + np.einsum("ia,jb->ijab", t1, t1)
    e = float(np.sum((2 * er.ovov.transpose(0, 2, 1, 3) - er.ovov.transpose(0, 2, 3, 1)) * tau))
    print(json.dumps(dict(s_per_iteration=dt, e_corr=e)), flush=True)


def parity_probe():
    """corr-E error vs the oracle on a small fragment of the same family (the metric's second half)."""
    sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
    from helpers import synthetic_fragment
    from qemb_oracle import ccsd, eri, scf
    from quemb_amd.fragsolver import DeviceFragment, default_opts
    n, o = 26, 6
    h, e1 = synthetic_fragment(n, o, SEED0)
    fr = DeviceFragment(n, 6); fr.set_eri_s4(eri.pack_s4(e1))
    out = fr.solve(o, h, opts=default_opts(), eeval=False)
    mf = scf.rhf(h, e1, o)
    _, _, ecc, _ = ccsd.solve_ccsd(h, e1, o, mf["mo_coeff"], mf["mo_energy"])
    fr.free()
    return abs(out["e_corr_mo"] - ecc)


def clock_probe(lib):
    """Sustained shader clock under the pp-ladder kernel: one launch of the ladder tile (224 x 128, one 8-wave workgroup per CU) on
    random operands with every workgroup recording its s_memtime ticks; sum(ticks) / (256 CUs x kernel time).  Outside the timed region."""
    from quemb_amd._lib import DeviceBuffer, check
    M, N, K = 210, 4096, 16384
    rng = np.random.default_rng(7)
    dA, dB, dC = DeviceBuffer.from_numpy(rng.standard_normal((M, K))), DeviceBuffer.from_numpy(rng.standard_normal((N, K))), DeviceBuffer(M * N)
    vals = []
    for _ in range(4):
        ms, ghz, wg = C.c_double(), C.c_double(), C.c_int64()
        check(lib.qemb_op_gemm_probe(M, N, K, dA.ptr, K, 1, dB.ptr, K, 1, dC.ptr, N, 13, 8, C.byref(ms), C.byref(ghz), C.byref(wg)), "qemb_op_gemm_probe", lib)
        vals.append((ghz.value, ms.value, wg.value))
    for b in (dA, dB, dC):
        b.free()
    ghz, ms, wg = sorted(vals[1:])[1]
    return dict(ghz=ghz, ms=ms, workgroups=wg, tflops=2.0 * M * N * K / (ms * 1e9))


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker, args.nocc, args.cpu_iters)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # QEMB_DIST_BACKEND=gloo + several ranks on one card is a rehearsal mode for 1-GPU boxes (RCCL refuses duplicate GPUs)
    backend = os.environ.get("QEMB_DIST_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    lrank = lrank % ndev
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(lrank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lrank))
        else:
            dist.init_process_group(backend=backend)
    from quemb_amd import _lib
    from quemb_amd.fragsolver import default_opts
    lib = _lib.init(lrank)
    n, o, F = args.n, args.nocc, args.frags_per_gpu
    v = n - o
    nf = min(22, n // 2)
    opts = default_opts()

    # ---- set-up (untimed): fragments resident in HBM, initial fragment SCF for dm0 (BE.initialize does the same)
    frs = []
    log(f"setting up {F} fragments per GPU (n={n}, n_occ={o})")
    for i in range(F):
        fr, h, _ = make_fragment(lib, n, nf, SEED0 + rank * F + i, args.scale)
        r = fr.scf(o, h, None, opts=opts)
        dm0 = 2.0 * r["mo_coeff"][:, :o] @ r["mo_coeff"][:, :o].T
        frs.append((fr, h, dm0))
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)

    comm_dev = torch.device("cuda", lrank) if backend == "nccl" else torch.device("cpu")
    buf_t = torch.zeros(8, dtype=torch.float64, device=comm_dev) if world > 1 else None

    nctx = 1
    pool = None
    if args.nstreams > 1:      # be_func(..., nstreams=k): worker threads, each bound to its own execution context (HIP stream)
        import queue
        from concurrent.futures import ThreadPoolExecutor
        from quemb_amd._lib import check
        nctx = lib.qemb_ctx_count(args.nstreams + 1)
        ids = queue.Queue()
        for k in range(1, args.nstreams + 1):
            ids.put(k)
        pool = ThreadPoolExecutor(max_workers=args.nstreams, initializer=lambda: check(lib.qemb_ctx_bind(ids.get()), "qemb_ctx_bind", lib))

    def one(t):
        fr, h, dm0 = t
        return fr.solve(o, h, dm0, opts=opts, eeval=True)

    def sweep():
        acc = np.zeros(8)
        for out in (pool.map(one, frs) if pool else map(one, frs)):
            acc[0] += out["n_iter"]; acc[1:4] += out["e_frag"]; acc[4] += np.trace(out["rdm1_emb"][:nf, :nf]); acc[5] += out["e_corr_mo"]; acc[6] += 1
        if world > 1:      # the one exchange of a sweep: residual/energy buffer, RCCL sum-all-reduce
            buf_t.copy_(torch.from_numpy(acc))
            dist.all_reduce(buf_t, op=dist.ReduceOp.SUM)
            acc = buf_t.cpu().numpy()
        return acc

    log("fragments resident; warm-up sweeps")
    for _ in range(args.warmup):
        sweep()
    log("timed sweeps")
    for s in range(8):
        for k in range(nctx):
            lib.qemb_ctx_timer_read(k, s, None, None, 1)
    if world > 1:
        dist.barrier()
    lib.qemb_sync(); sync()
    t0 = time.perf_counter()
    tot = np.zeros(8)
    for _ in range(args.steps):
        tot += sweep()
    lib.qemb_sync(); sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    else:
        pass
    # `tot` is already summed over ranks inside sweep() when world > 1
    n_iter_total = float(tot[0]); n_frag_total = float(tot[6])

    log(f"timed region done: {dt:.2f} s for {args.steps} step(s)")
    if rank == 0:
        lad_ms, lad_cnt = read_timer(lib, 0, nctx)
        it_ms, it_cnt = read_timer(lib, 2, nctx)
        ao_ms, ao_cnt = read_timer(lib, 3, nctx)
        scf_ms, scf_cnt = read_timer(lib, 4, nctx)
        ring_ms, ring_cnt = read_timer(lib, 1, nctx)
        lad_avg = lad_ms / max(lad_cnt, 1) * 1e-3
        npair_o = o * (o + 1) // 2
        npv, nmv, nmo = v * (v + 1) // 2, v * (v - 1) // 2, o * (o - 1) // 2
        flop_ladder = 2.0 * npair_o * float(npv) ** 2 + 2.0 * nmo * float(nmv) ** 2   # executed: (+/-) pair-packed products
        flop_dense = 2.0 * o * o * float(v) ** 4                 # SURVEY 8(d) dense-equivalent figure
        achieved = flop_ladder / lad_avg / 1e12 if lad_avg > 0 else 0.0
        traffic = None                                           # HBM bytes per launch from the separate --pmc passes
        pmc = ROOT / "profiles" / "r01_pmc_ladder.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        res = {
            "metric": "fragment CCSD iters/sec (full BE sweep over synthetic n_occ=20 n_virt=200 fragments); corr-E error vs oracle in parity_max_abs_err_Eh",
            "value": n_iter_total / dt, "unit": "CCSD iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: synthetic fragment sweep, {F} fragments per GPU ({F * world} total), "
                                   f"n_occ={o} n_virt={v} (n={n}), DF-factorised ERIs naux={3 * n} scale={args.scale}, "
                                   "one be_func sweep per step (fragment RHF + MO transform + RCCSD to |dE|<1e-10 + energies + 1 all-reduce)",
                       "fragments_per_gpu": F, "n_occ": o, "n_virt": v, "fragments_in_flight_per_gpu": args.nstreams, "parallelism": f"fragments sharded over {world} GPU(s), 1 RCCL all-reduce per sweep"},
            "fragments_per_s": n_frag_total / dt,
            "ccsd_iterations_per_fragment": n_iter_total / max(n_frag_total, 1),
            "mean_e_corr_per_fragment": float(tot[5]) / max(n_frag_total, 1),
            "roofline": {"bound": "mfma", "kernel": "dgemm_mfma_kernel<7,2,2,4,16,true,true,2,1> (+ pairs, 224x128 tile) and <6,2,2,4,16,true,true,2,1> (- pairs, 192x128): pp-ladder over (+/-) packed pairs, M=npair(o) N=K=npair(v), split-K + slab reduce",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                         # one pp-ladder = TWO dispatches of this kernel ((+) and (-) pair blocks); per-dispatch averages:
                         "traffic": None if traffic is None else traffic / 2.0, "avg_launch_ms": lad_avg * 1e3 / 2.0,
                         "launches": 2 * lad_cnt, "flop_per_launch": flop_ladder / 2.0, "ladder_ms": lad_avg * 1e3,
                         "dense_equivalent_tflops": flop_dense / lad_avg / 1e12 if lad_avg > 0 else 0.0,
                         "algorithmic_bytes_per_launch": 4.0 * (float(npv) ** 2 + float(nmv) ** 2 + 2.0 * npair_o * npv + 2.0 * nmo * nmv)},
            "device_time_ms_rank0": {"ccsd_iteration_avg": it_ms / max(it_cnt, 1), "ccsd_iterations": it_cnt, "rings_avg": ring_ms / max(ring_cnt, 1),
                                     "mo_transform_avg": ao_ms / max(ao_cnt, 1), "fragment_scf_avg": scf_ms / max(scf_cnt, 1)},
        }
        if world == 1:
            try:
                cp = clock_probe(lib)
                clk = cp["ghz"]
                res["roofline"].update({"sustained_clock_ghz": clk, "peak_at_sustained_clock": PEAK_FP64_MFMA_TFLOPS * clk / 2.4,
                                        "frac_of_sustained_peak": achieved / (PEAK_FP64_MFMA_TFLOPS * clk / 2.4) if clk > 0 else None,
                                        "clock_probe_tflops": cp["tflops"],
                                        "clock_probe": "launches of the ladder tile (cfg 13 = the same code under the untagged kernel symbol, M=210 N=4096 K=16384, split-K 8 = 256 workgroups, random operands) "
                                                       "right after the timed region, GPU still hot: sum of per-workgroup s_memtime ticks / (256 CUs x kernel time)"})
            except Exception as e:  # noqa: BLE001
                res["roofline"]["sustained_clock_ghz"] = f"probe failed: {e}"
            log("parity probe vs oracle")
            try:
                res["parity_max_abs_err_Eh"] = parity_probe()
            except Exception as e:  # noqa: BLE001
                res["parity_max_abs_err_Eh"] = f"probe failed: {e}"
            if not args.no_cpu_baseline:
                fr0, h0, dm00 = frs[0]
                res["cpu_baseline"] = cpu_baseline(fr0, h0, dm00, o, opts, args.cpu_iters, args.cpu_threads)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
