"""From a rocprofv3 --kernel-trace CSV of tools/octane_lockstep.py: the grouped launches of the lock-step iterations -- count, mean duration and
mean gap to the previous kernel -- and the same for the ungrouped kernels (profiles/r03_octane_lockstep_trace.txt)."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = {}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"]
    grouped = "grouped_kernel" in nm
    key = "grouped" if grouped else "single"
    short = nm
    if grouped:
        i = nm.find("_ZNS_"); short = "grouped:" + nm[i + 5:i + 45]
    a = agg.setdefault(short[:70], [0, 0.0, 0.0, key])
    a[0] += 1; a[1] += (e - s) / 1e3
    if prev_end is not None and s - prev_end < 200000: a[2] += max(0, s - prev_end) / 1e3
    prev_end = e
tot = {"grouped": [0, 0.0, 0.0], "single": [0, 0.0, 0.0]}
for k, (c, d, g, key) in agg.items():
    tot[key][0] += c; tot[key][1] += d; tot[key][2] += g
for key, (c, d, g) in tot.items():
    print(f"{key:8s}: {c:6d} launches, mean duration {d / max(c, 1):6.2f} us, mean gap before {g / max(c, 1):6.2f} us")
for k, (c, d, g, key) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{c:6d}x  dur {d / c:7.2f} us  gap {g / c:6.2f} us  {k}")
