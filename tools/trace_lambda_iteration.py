"""From a rocprofv3 --kernel-trace CSV of tools/lambda_bench.py: per-kernel time of ONE Lambda iteration (two applications of the
pp-ladder per iteration: the window between the 3rd-last and the last (+)-ladder launches of the relaxed solve)."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lad = [i for i, r in enumerate(rows) if "<7, 2, 2, 4, 16, true, true, 2, 1" in r["Kernel_Name"]]
# the densities pass after the last iteration also applies the ladder: step back a few launches to sit inside the iterations
a, b = lad[-7], lad[-5]
agg = {}
busy = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("qemb::", "").replace("void ", "").split("(")[0][:80] + "  grid " + str(r.get("Grid_Size", "?"))
    k = agg.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += d
    busy += d
tot = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print(f"Lambda iteration {tot:.1f} us, kernels busy {busy:.1f} us, {b - a} kernels")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{d:9.1f} us  {c:4d}x  {k}")
