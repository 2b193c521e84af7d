"""From a rocprofv3 --kernel-trace CSV of tools/octane_lockstep.py: the LAST sweep split at its first and last grouped ph_layouts launch -- kernels before the
lock-step iterations (fragment RHF, MO integrals, set-up), the iterations, kernels after them (RDMs, energies) -- with launch count, summed kernel time and wall span
of each part, and the ten kernels that carry the most time before and after the iterations."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
it = [i for i, r in enumerate(rows) if "ccsd_ph_layouts" in r["Kernel_Name"] and "grouped" in r["Kernel_Name"]]
# sweeps: runs of iterations separated by more than 1 ms without a grouped ph_layouts launch
sweeps = [[it[0]]]
for a, b in zip(it, it[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 2_500_000: sweeps.append([])
    sweeps[-1].append(b)
last, prev = sweeps[-1], sweeps[-2]
def part(lo, hi, label):
    seg = rows[lo:hi]
    if not seg: return
    k = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    print(f"{label}: {len(seg)} launches, {k:.0f} us in kernels, {span:.0f} us wall span")
    agg = {}
    for r in seg:
        nm = r["Kernel_Name"]
        i = nm.find("_ZNS_")
        nm = ("grouped:" + nm[i + 5:i + 60]) if "grouped" in nm and i >= 0 else nm[:70]
        a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"    {t:8.1f} us  {c:4d} x  {nm}")
# the end of the previous sweep's post phase = the last kernel before a gap > 300 us ahead of this sweep's first iteration
first_it, last_it = last[0], last[-1]
lo = prev[-1]
# walk forward from the previous sweep's last iteration to find the largest gap: that is the sweep boundary
best, cut = -1, lo
for i in range(lo, first_it):
    g = int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])
    if g > best: best, cut = g, i + 1
part(cut, first_it, "before the iterations")
part(first_it, last_it, "iterations (all but the last)")
part(last_it, len(rows), "last iteration + after the iterations")
