#!/bin/bash
# PMC view of the GEMM main-loop variants (run on the GPU box through gpurun): SQ counters in one pass, GRBM_GUI_ACTIVE (effective
# clock = count / 8 XCDs / kernel time, MI355X guide "DVFS give-back") in another.  Writes gpurun_out/gemm_pmc.json.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_grbm
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq -- python tools/gemm_modes.py 3 ${QEMB_PMC_SET:-pmc} > gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -- python tools/gemm_modes.py 3 ${QEMB_PMC_SET:-pmc} > gpurun_out/pmc_grbm.log 2>&1
python - <<'PY'
import csv, glob, json, collections
out = collections.OrderedDict()
def load(d):
    cc = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0))))
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return dur, vals
for d in ("pmc_sq", "pmc_grbm"):
    dur, vals = load(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for did, cs in vals.items():
        ns, name, grid = dur.get(did, (0, "?", 0))
        if "dgemm_mfma_kernel" not in name or ns < 300000:
            continue
        key = name.replace("void qemb::", "").replace("(qemb::GemmKArgs)", "") + f" grid={grid} #{len([k for k in agg if k.startswith(name[:10])])}" if False else name.replace("void qemb::", "").replace("(qemb::GemmKArgs)", "") + f" grid={grid}"
        agg[key]["ns"].append(ns)
        for c, v in cs.items():
            agg[key][c].append(v)
    for key, cs in agg.items():
        o = out.setdefault(key, {})
        for c, v in cs.items():
            o[c if c != "ns" else f"ns_{d}"] = sum(v) / len(v)
        o[f"dispatches_{d}"] = len(cs["ns"])
for key, o in out.items():
    if "GRBM_GUI_ACTIVE" in o:
        o["effective_clock_ghz"] = o["GRBM_GUI_ACTIVE"] / 8.0 / o["ns_pmc_grbm"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "ns_pmc_sq" in o:
        # busy cycles summed over the 1024 SIMDs
        o["mfma_busy_cycles_per_simd_per_us"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (o["ns_pmc_sq"] / 1e3)
    if o.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_conflict_frac"] = o.get("SQ_LDS_BANK_CONFLICT", 0.0) / o["SQ_LDS_IDX_ACTIVE"]
    if o.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            o[c + "_frac_of_wave_cycles"] = o.get(c, 0.0) / o["SQ_WAVE_CYCLES"]
json.dump(out, open("gpurun_out/gemm_pmc.json", "w"), indent=1)
for key, o in out.items():
    print(key)
    print("   ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o.items() if "frac" in k or "clock" in k or "per_us" in k or k.startswith("ns_")})
PY
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_grbm
