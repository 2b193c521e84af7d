"""From a rocprofv3 --kernel-trace CSV of tools/frag_bench.py: the ordered kernel list of the LAST CCSD iteration (from one
amplitude-layout pass -- the first kernel of update_amps -- to the next), with durations and the idle gaps between kernels.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20
    python tools/trace_iteration.py gpurun_out/kt
"""
import csv, glob, sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lad = [i for i, r in enumerate(rows) if "ccsd_ph_layouts_kernel" in r["Kernel_Name"]]
a, b = lad[-2], lad[-1]
t_prev = None
tot = busy = 0.0
agg = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t_prev) / 1e3 if t_prev else 0.0
    name = r["Kernel_Name"].replace("qemb::", "").replace("void ", "")
    short = name.split("(")[0][:70]
    d = (e - s) / 1e3
    print(f"{d:9.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  {short}")
    busy += d
    k = agg.setdefault(short, [0, 0.0]); k[0] += 1; k[1] += d
    t_prev = e
tot = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print(f"\niteration {tot:.1f} us, kernels busy {busy:.1f} us, idle {tot - busy:.1f} us, {b - a} kernels")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us  {c:4d}x  {k}")
