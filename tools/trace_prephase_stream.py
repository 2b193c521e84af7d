"""From a rocprofv3 --kernel-trace CSV of tools/octane_lockstep.py: the kernels ONE fragment's stream runs before the lock-step iterations of the last sweep (fragment RHF,
MO integrals, CCSD set-up), in launch order with start time, duration and gap -- and the same for the phase after the iterations with argument `after`."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
it = [i for i, r in enumerate(rows) if "ccsd_ph_layouts" in r["Kernel_Name"] and "grouped" in r["Kernel_Name"]]
sw = [it[0]]
for a, b in zip(it, it[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 2_500_000: sw.append(b)
first = sw[-1]; prev_last = [i for i in it if i < first][-1]
best, cut = -1, prev_last
for i in range(prev_last, first):
    g = int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])
    if g > best: best, cut = g, i + 1
seg = rows[cut:first] if len(sys.argv) < 3 else rows[it[-1]:]
qs = {}
for r in seg: qs.setdefault(r["Queue_Id"], []).append(r)
print({q: len(v) for q, v in qs.items()})
q = sorted(qs, key=lambda k: len(qs[k]))[-1 if len(sys.argv) < 3 else -2]
t0 = int(seg[0]["Start_Timestamp"])
prev = None
for r in qs[q]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("qemb::", "").replace("void ", "").split("(")[0][:70]
    print("%8.1f  %6.1f us gap %6.1f  grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, r["Grid_Size_X"], nm))
    prev = e
