"""From a rocprofv3 --kernel-trace CSV of tools/frag_bench.py: the LAST fragment solve split into its phases -- everything before the
first pp-ladder launch (fragment RHF, MO transformation, CCSD setup), the CCSD iterations, everything after the last one (1-RDM,
energies) -- with the kernels of the two non-iteration phases listed by total time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20
    python tools/trace_solve.py gpurun_out/kt
"""
import csv, glob, sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("qemb::", "").replace("void ", "").split("(")[0][:90]
lad = [i for i, r in enumerate(rows) if "<7, 2, 2, 4, 16, true, true, 2, 1" in r["Kernel_Name"]]
# solves are separated by long runs without a ladder launch: split the ladder indices at gaps > 50 kernels... use time gaps instead
starts = [int(rows[i]["Start_Timestamp"]) for i in lad]
per = sorted(b - a for a, b in zip(starts, starts[1:]))
it_ns = per[len(per) // 2]
cut = max(k for k in range(1, len(lad)) if starts[k] - starts[k - 1] > 2.5 * it_ns)        # first ladder of the last solve
prev_last = lad[cut - 1]
first, last = lad[cut], lad[-1]
# the unpack of the packed ERIs into pair rows opens a solve
open_idx = min(i for i in range(prev_last, first) if "unpack_tril" in rows[i]["Kernel_Name"])

def phase(label, a, b):
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b - 1]["End_Timestamp"])
    agg, busy = {}, 0.0
    for r in rows[a:b]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += d
        k = agg.setdefault(name(r), [0, 0.0]); k[0] += 1; k[1] += d
    print(f"\n== {label}: wall {(t1 - t0) / 1e6:.2f} ms, kernels busy {busy / 1e3:.2f} ms, {b - a} launches")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"{d / 1e3:9.3f} ms  {c:5d}x  {k}")


phase("before the first ladder launch (RHF + MO transformation + CCSD setup + start of iteration 1)", open_idx, first)
print(f"\n== CCSD iterations: {(int(rows[last]['Start_Timestamp']) - int(rows[first]['Start_Timestamp'])) / 1e6:.2f} ms between the first and the last ladder launch, "
      f"{len(lad) - cut} ladder launches, median period {it_ns / 1e6:.3f} ms")
phase("after the last ladder launch (rest of the last iteration, 1-RDM, energies)", last, len(rows))
