"""From a rocprofv3 --kernel-trace CSV of tools/octane_lockstep.py: the kernels of ONE lock-step iteration in launch order (from one grouped ph_layouts launch to
the next), with duration, gap and grid -- which of the grouped launches carry the time."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "grouped_kernel" in r["Kernel_Name"] and "ccsd_ph_layouts" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
prev = None
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"]
    if "grouped_kernel" in nm:
        i = nm.find("_ZNS_"); nm = "grouped:" + nm[i + 5:i + 75]
    gap = 0.0 if prev is None else (s - prev) / 1e3
    print(f"{(e - s) / 1e3:8.2f} us  gap {gap:6.2f}  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} wg  {nm[:100]}")
    tot += (e - s) / 1e3
    prev = e
print(f"iteration: {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us wall, {tot:.1f} us in {b - a} kernels")
