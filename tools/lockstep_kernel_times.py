"""From a rocprofv3 --kernel-trace CSV of tools/octane_lockstep.py: mean duration of every kernel that ran as a grouped launch, by body name
(per lock-step iteration of the six octane fragments), sorted by total time.  A/B aid for the small-fragment kernel variants."""
import csv, glob, re, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
agg = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if "grouped_kernel" not in nm:
        continue
    m = re.search(r"_ZNS_\d+([A-Za-z0-9_]+?)_bodyE?", nm) or re.search(r"_ZNS_\d+([A-Za-z0-9_]+)", nm)
    key = (m.group(1) if m else nm[:60])
    if "dgemm" in key:
        t = re.search(r"dgemm_mfma_bodyILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)", nm)
        key = "dgemm<%s>" % ",".join(t.groups()) if t else key
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / len(v):8.2f} us x {len(v):5d}   {k}")
    tot += sum(v)
n_it = max(1, len(agg.get("ccsd_ph_layouts_kernel", agg.get("ccsd_ph_layouts_small_kernel", [1]))))
print(f"grouped kernels: {tot / n_it:.1f} us per lock-step iteration ({n_it} iterations)")
