"""From a rocprofv3 --hip-trace CSV: HIP API calls that took longer than a threshold (ms), in time order."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)
thr_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > thr_ms:
        print(f"t={(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  {d:9.2f} ms  {r.get('Function', r.get('Name', '?'))}  thread {r.get('Thread_Id', '?')}")
print(len(rows), "calls")
