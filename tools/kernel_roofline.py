"""Per-kernel roofline table of one fragment solve (SURVEY 8d "reported results") from a rocprofv3 --kernel-trace CSV of tools/frag_bench.py:
the MFMA-bound products against the 78.6 TFLOP/s FP64 matrix peak, the HBM-bound passes against 8 TB/s (algorithmic bytes: every operand read
or written once).

    QEMB_GEMM_SHAPELOG=gpurun_out/shapes.txt rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 eeval
    python tools/kernel_roofline.py "gpurun_out/kt/*/*kernel_trace.csv" gpurun_out/shapes.txt [n n_occ naux]

Products are identified BY SHAPE (round 5): the library writes M N K batch of every product in launch order (QEMB_GEMM_SHAPELOG), the i-th
dgemm_mfma_kernel dispatch of the single-stream trace is the i-th line, and flops = 2 M N K batch of that very call.  (Round 4 keyed the flop count on the
kernel's template string: under the factor route other shapes ran on the same tiles and the table showed 20 x the peak.)  Rows are grouped by
(kernel symbol, shape); `what` names the shapes of the solve that are recognised, anything else is listed as it is.
HBM passes are still keyed by kernel name with the bytes of their large call at (n, n_occ); a row above its peak is an error of this table and fails
tests/test_boundary_docs.py.
"""
import csv, glob, json, sys

PEAK_F, PEAK_B = 78.6e12, 8.0e12


def describe(M, N, K, batch, n, o, naux):
    v = n - o
    npn, npo, nmo, npv, nmv, nov = n * (n + 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2, o * v
    ldp, ldm = npv + (npv & 1), max(2, nmv + (nmv & 1))
    known = {
        (npo, npv, ldp, 1): "pp-ladder, (+) packed pairs",
        (nmo, nmv, ldm, 1): "pp-ladder, (-) packed pairs",
        (nov, nov, nov, 1): "ph ring product (ov)^3",
        (npo, nov, ldp, 1): "tau-side dressing of W_vvvv, (+)",
        (nmo, nov, ldm, 1): "tau-side dressing, (-)",
        (naux * n, n, n, 1): "factor route: Lh = B C (tall product over (L,p) rows)",
        (n, n, n, naux): "factor route: Lmo = C^T Lh, batched over the auxiliary index",
        (npn, nf_cols(n), naux, 1): "factor route: T = Lpk^T Lh (3/4-transformed integrals of the energies, K = naux)",
        (n, npn * n, n, 1): "four-index route: first quarter transform",
        (npn * n, n, n, 1): "four-index route: slab . C as one tall product",
        (n, n, n, npn): "four-index route: C^T . slab batched over the pairs",
        (o * o * v, v, v, 1): "U = t2 . Lvv'",
        (o * v * v, o, v, 1): "ZB = ovvv . t1",
        (v, v, o * o * v, 1): "Fvv' = tau . Loovv (long K, split)",
        (npo, o * o, ldp, 1): "Woooo += ovov . tau over (+) packed pairs",
        (nmo, o * o, ldm, 1): "Woooo += ovov . tau over (-) packed pairs",
        (npo, npv, npo, 1): "hole-hole ladder, (+)",
        (nmo, nmv, nmo, 1): "hole-hole ladder, (-)",
        (o, n, n, naux): "fragment RHF exchange from the factor: Y_L = Co^T B_L",
        (n, n, naux * o, 1): "fragment RHF exchange from the factor: K = 2 Y^T Y (long K, split)",
        (n, n, n, 1): "n x n x n (fragment RHF: FD, rotations)",
    }
    if (M, N, K, batch) in known:
        return known[(M, N, K, batch)]
    if K == naux and batch == 1 and M <= npn and N <= npn and M + N > npn // 2:
        return "factor route: block column of Lpk^T Lpk (K = naux)"
    return None


def nf_cols(n):
    return min(22, n // 2) * n


def bytes_table(n, o):
    """kernel-name fragment -> (algorithmic bytes of one large call, what) at fragment size (n, n_occ)"""
    v = n - o
    npn, npo, nmo, npv, nmv = n * (n + 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2
    N2 = o * o * v * v * 8
    bytes_of = {   # kernel-name fragment -> (algorithmic bytes of one large call, what)
        "unpack_tril_tiled_kernel": ((npn * npn + npn * n * n) * 8, "packed rows -> unpacked n x n slabs (read packed + write unpacked)"),
        "jk_packed_rowgroups": (npn * npn * 8, "Coulomb + exchange matrix, one pass over the 4-fold packed block"),
        "pack_pm_tiled_kernel<0>": ((o * v * v * v + o * v * (npv + nmv)) * 8, "(+/-) packed images of the ovvv block through LDS tiles"),
        "pack_pm_tiled_kernel<1>": ((npo * v * v + npo * npv + nmo * nmv) * 8, "tau -> (+/-) packed pair rows through LDS tiles"),
        "ladder_pack_vvvv_pf_kernel": (2 * (npv * npv + nmv * nmv) * 8, "(+/-) ladder operands from the pair-first MO tensor"),
        "ladder_scatter_pm_kernel": (N2 + (8 + 1) * (npo * npv + nmo * nmv) * 8, "ladder result -> t2: the 8 split-K slabs of R+/R- added on the way, + the hole-hole rows, t2 written"),
        "ccsd_ph_layouts_kernel": (7 * N2, "t2 -> T, T', u, u~, T'~, Theta layouts in one pass"),
        "ccsd_finish_t2_rings_kernel": (6 * N2, "(t2n + ovov + U' + U'^T) / D with the two ring products read where the GEMMs leave them"),
        "diis_push_kernel": (8 * N2, "DIIS push in one pass: e = t_new - t, Gram row against the five older error vectors"),
        "ccsd_extrapolate_energy_kernel": (9 * N2, "DIIS extrapolation (six vectors in), tau and the energy reduction in one pass"),
        "small_k_update": (2 * N2, "rank-n_occ update of an o^2 v^2 tensor (r/w), MFMA"),
        "mirror_lower_kernel": (npn * npn * 8, "mirror of the lower block columns of the pair product (r/w of half the block each)"),
    }
    return bytes_of


bytes_of = bytes_table(220, 20)      # (tools/hbm_pmc.py reads the table of the benchmarked size)


def main():
    args = [a for a in sys.argv[1:]]
    files = [f for f in glob.glob(args[0])]
    shapes = [tuple(int(x) for x in ln.split()) for ln in open(args[1])] if len(args) > 1 else None
    n, o, naux = (int(args[2]), int(args[3]), int(args[4])) if len(args) > 4 else (220, 20, 660)
    bytes_of = bytes_table(n, o)
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gemm_rows = [r for r in rows if "dgemm_mfma_kernel<" in r["Kernel_Name"]]
    out = []
    if shapes is not None:
        if len(shapes) != len(gemm_rows):
            raise SystemExit(f"{len(gemm_rows)} dgemm_mfma_kernel dispatches in the trace, {len(shapes)} lines in the shape log: not the same run (or not single-stream)")
        agg = {}
        for r, sh in zip(gemm_rows, shapes):
            sym = r["Kernel_Name"][r["Kernel_Name"].index("dgemm_mfma_kernel"):].split("(")[0]
            agg.setdefault((sym, sh[:4]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
        for (sym, (M, N, K, batch)), ds in agg.items():
            f = 2.0 * M * N * K * batch
            if f < 2.0e10:
                continue                                     # the table lists the large products
            t = sum(ds) / len(ds)
            out.append(dict(kernel=sym, bound="mfma", M=M, N=N, K=K, batch=batch, what=describe(M, N, K, batch, n, o, naux) or f"{M} x {N} x {K}" + (f", batch {batch}" if batch > 1 else ""),
                            calls=len(ds), mean_ms=round(t * 1e3, 4), total_ms=round(sum(ds) * 1e3, 3), executed_GFLOP=round(f / 1e9, 1), TFLOPs=round(f / t / 1e12, 2),
                            frac_of_78_6_TFLOPs=round(f / t / PEAK_F, 3)))
    agg = {}
    for r in rows:
        for key in bytes_of:
            if key in r["Kernel_Name"]:
                agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for key, ds in agg.items():
        top = [d for d in ds if d >= 0.5 * max(ds)]
        t = sum(top) / len(top)
        b, what = bytes_of[key]
        out.append(dict(kernel=key, bound="hbm", what=what, calls=len(ds), large_calls=len(top), mean_ms=round(t * 1e3, 4), algorithmic_GB=round(b / 1e9, 3),
                        TBps=round(b / t / 1e12, 2), frac_of_8_TBps=round(b / t / PEAK_B, 3)))
    out.sort(key=lambda d: -(d.get("total_ms") or d["mean_ms"] * d.get("large_calls", 1)))
    for d in out:
        print(json.dumps(d))


if __name__ == "__main__":
    main()
