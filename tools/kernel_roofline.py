"""Per-kernel roofline table of one fragment solve (n = 220, n_occ = 20; SURVEY 8d "reported results") from a rocprofv3
--kernel-trace CSV of `python tools/frag_bench.py 220 20`: the MFMA-bound products against the 78.6 TFLOP/s FP64 matrix peak (executed
flops of the call shape), the HBM-bound passes against 8 TB/s (algorithmic bytes: every operand read or written once).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20
    python tools/kernel_roofline.py gpurun_out/kt/*/*kernel_trace.csv

Kernels that share a symbol across call sites of different sizes (the generic copy / GEMM instantiations) are represented by their
large calls: the mean over the calls within 2x of the longest one.
"""
import csv, glob, json, sys

n, o = 220, 20
v = n - o
npn, npo, nmo, npv, nmv, nov = n * (n + 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2, o * v
N2 = o * o * v * v * 8
PEAK_F, PEAK_B = 78.6e12, 8.0e12

flops_of = {   # kernel-symbol fragment -> (executed flop of one large call, what)
    "<7, 2, 2, 4, 16, true, true, 2, 1, 1>": (2.0 * npo * npv * npv, "pp-ladder, (+) packed pairs: 210 x 20100 x 20100"),
    "<6, 2, 2, 4, 16, true, true, 2, 1, 1>": (2.0 * nmo * nmv * nmv, "pp-ladder, (-) packed pairs: 190 x 19900 x 19900"),
    "<4, 4, 2, 4, 16, true, true, 2, 0, 1>": (2.0 * nov ** 3, "ph ring product (ov)^3 = 4000^3"),
    "<7, 2, 2, 4, 16, true, true, 2, 0, 1>": (2.0 * npo * nov * npv, "tau-side dressing of W_vvvv, (+): 210 x 4000 x 20100"),
    "<6, 2, 2, 4, 16, true, true, 2, 0, 1>": (2.0 * nmo * nov * nmv, "tau-side dressing, (-): 190 x 4000 x 19900"),
    "<7, 2, 2, 4, 16, false, true, 2, 0, 1>": (2.0 * n * n * npn * n, "MO transformation, first quarter transform: 220 x (24310 * 220) x 220"),
    "<2, 7, 4, 2, 16, true, false, 2, 0, 1>": (2.0 * n * n * npn * n, "MO transformation, slab . C flat: (24310 * 220) x 220 x 220"),
    "<7, 2, 2, 4, 16, false, false, 2, 0, 1>": (2.0 * n * n * npn * n, "MO transformation, C^T . slab batched over 24310 pairs"),
}
bytes_of = {   # kernel-name fragment -> (algorithmic bytes of one large call, what)
    "unpack_tril_tiled_kernel": ((npn * npn + npn * n * n) * 8, "packed rows -> unpacked n x n slabs (read packed + write unpacked)"),
    "jk_packed_stage1": (npn * npn * 8, "Coulomb + exchange matrix, one pass over the 4-fold packed block"),
    "jk_packed_rowgroups": (npn * npn * 8, "Coulomb + exchange matrix, one pass over the 4-fold packed block (round 3: 16-lane row groups, 16-byte loads)"),
    "pack_pm_tiled_kernel<0>": ((o * v * v * v + o * v * (npv + nmv)) * 8, "(+/-) packed images of the ovvv block through LDS tiles (round 3)"),
    "pack_pm_tiled_kernel<1>": ((npo * v * v + npo * npv + nmo * nmv) * 8, "tau -> (+/-) packed pair rows through LDS tiles (round 3)"),
    "ladder_pack_vvvv_pf_kernel": (2 * (npv * npv + nmv * nmv) * 8, "(+/-) ladder operands from the pair-first MO tensor"),
    "pack_pm_cols_kernel": ((o * v * v * v + o * v * (npv + nmv)) * 8, "(+/-) packed images of the ovvv block"),
    "ladder_scatter_pm_kernel": (N2 + (8 + 1) * (npo * npv + nmo * nmv) * 8, "ladder result -> t2: the 8 split-K slabs of R+/R- added on the way, + the hole-hole rows, t2 written (round 3: no reduction pass)"),
    "ladder_pack_tau_kernel": (N2 + (npo * npv + nmo * nmv) * 8, "tau -> (+/-) packed pair rows"),
    "ccsd_ph_layouts_kernel": (7 * N2, "t2 -> T, T', u, u~, T'~, Theta layouts in one pass"),
    "ccsd_finish_t2_rings_kernel": (6 * N2, "(t2n + ovov + U' + U'^T) / D with the two ring products read where the GEMMs leave them, each (i >= j) pair of tiles once (round 3)"),
    "diis_push_kernel": (8 * N2, "DIIS push in one pass: e = t_new - t, Gram row against the five older error vectors (round 3)"),
    "ccsd_extrapolate_energy_kernel": (9 * N2, "DIIS extrapolation (six vectors in), tau and the energy reduction in one pass (round 3)"),
    "small_k_update": (2 * N2, "rank-n_occ update of an o^2 v^2 tensor (r/w), MFMA"),
}

def main():
    files = [f for a in sys.argv[1:] for f in glob.glob(a)]
    rows = list(csv.DictReader(open(files[0])))
    agg = {}
    for r in rows:
        for key in list(flops_of) + list(bytes_of):
            if key in r["Kernel_Name"]:
                agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for key, ds in agg.items():
        top = [d for d in ds if d >= 0.5 * max(ds)]
        t = sum(top) / len(top)
        if key in flops_of:
            f, what = flops_of[key]
            print(json.dumps(dict(kernel="dgemm_mfma_kernel" + key, bound="mfma", what=what, calls=len(ds), large_calls=len(top), mean_ms=round(t * 1e3, 4),
                                  executed_GFLOP=round(f / 1e9, 1), TFLOPs=round(f / t / 1e12, 2), frac_of_78_6_TFLOPs=round(f / t / PEAK_F, 3))))
        else:
            b, what = bytes_of[key]
            print(json.dumps(dict(kernel=key, bound="hbm", what=what, calls=len(ds), large_calls=len(top), mean_ms=round(t * 1e3, 4),
                                  algorithmic_GB=round(b / 1e9, 3), TBps=round(b / t / 1e12, 2), frac_of_8_TBps=round(b / t / PEAK_B, 3))))


if __name__ == "__main__":
    main()
