"""From a rocprofv3 --kernel-trace CSV: device idle gaps above a threshold and kernels above a threshold, in time order."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
thr_us = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and (s - end) / 1e3 > thr_us:
        print(f"t={(end - t0) / 1e6:10.2f} ms  GAP {(s - end) / 1e3:10.1f} us before  {r['Kernel_Name'][:110]}")
    if (e - s) / 1e3 > thr_us:
        print(f"t={(s - t0) / 1e6:10.2f} ms  KERNEL {(e - s) / 1e3:10.1f} us  {r['Kernel_Name'][:110]}")
    end = e if end is None else max(end, e)
print(f"{len(rows)} kernels, span {(end - t0) / 1e6:.1f} ms")
