"""HBM-bound kernels of one fragment solve (n = 220, o = 20) against the 8 TB/s roofline, from a rocprofv3 --kernel-trace CSV of
`python tools/frag_bench.py 220 20`.  Algorithmic bytes are the operand sizes each kernel must read + write once."""
import csv, json, re, sys
n, o = 220, 20
v = n - o
npn, npv = n * (n + 1) // 2, v * (v + 1) // 2
N2 = o * o * v * v * 8
GB = 1e9
bytes_of = {   # kernel-name fragment -> (algorithmic bytes, what)
    "unpack_tril_tiled_kernel": ((npn * npn + npn * n * n) * 8, "s4 rows -> [P(pq)][r][s] (read packed + write unpacked)"),
    "k_pairs_stage1": (npn * n * n * 8, "exchange matrix from the half-unpacked tensor (read once)"),
    "pack_pair_rows_kernel": (2 * npn * npn * 8, "keep r' >= s' rows"),
    "ladder_pack_vvvv_pf_kernel": (2 * (npv * npv + (npv - v) ** 2) * 8, "(+/-) ladder operands from the pair-first MO tensor"),
    "ladder_scatter_pm_kernel": (2 * N2 + 2 * (o * (o + 1) // 2) * npv * 8, "ladder result -> four index images of t2 (r/w t2 + read R+/R-)"),
    "lincomb_kernel": (3 * N2, "a x + b y (two reads + one write of an o^2 v^2 tensor)"),
    "div_denom_kernel": (2 * N2, "t2 / D"),
}
rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    for key in bytes_of:
        if key in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            agg.setdefault(key, []).append(d)
for key, ds in agg.items():
    # calls on small tensors (t1-sized, single rows) share the kernel name: average the calls within 2x of the longest one
    top = [d for d in ds if d >= 0.5 * max(ds)]
    t = sum(top) / len(top)
    b, what = bytes_of[key]
    print(json.dumps(dict(kernel=key, what=what, calls=len(ds), mean_ms_large_calls=t * 1e3, algorithmic_GB=b / GB, GBps=b / t / GB,
                          frac_of_8TBps=b / t / 8e12)))
