"""Issue-side account of the HBM-bound kernels of one n = 220, n_occ = 20 fragment solve (tools/hbm_issue_pmc.sh).

    python tools/hbm_issue_pmc.py <directory of a counter pass> ...  > profiles/rNN_hbm_issue_pmc.json

Per kernel symbol of tools/kernel_roofline.py's HBM table (large calls only: within 2 x of the longest of the symbol), the mean of every counter and
what they say about WHY a pass is short of 8 TB/s:
  wait_frac        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   share of a wave's life spent waiting for an outstanding instruction (memory, for these kernels)
  issue_frac       SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES share spent issuing
  waves_per_cu     4 x SQ_WAVE_CYCLES / shader cycles / 256 CUs   waves resident on a CU on average (the SQ counters count quad-cycles; shader cycles =
                   GRBM_GUI_ACTIVE / 8 XCDs of the same dispatch in its own pass)
  vmem_per_wave    (SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR) / SQ_WAVES
  bytes_per_lane   algorithmic bytes / (64 x VMEM instructions): 8 = one double per lane and instruction, 16 = dwordx4 accesses
  valu_per_vmem, salu_per_vmem   address arithmetic and control per memory instruction
  vmem_latency_cycles  the derived counter VmemLatency (accumulated SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM): mean cycles a memory instruction is outstanding
  vmem_in_flight_per_cu  Little's law: VMEM instructions x latency / shader cycles of the kernel / 256 CUs
  bytes_in_flight_per_cu  vmem_in_flight_per_cu x 64 x bytes_per_lane; the rate these bytes sustain is bytes_in_flight x 256 / latency
  tcp_pending_stall_frac  TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1 (vector cache stalled on its pending-request FIFO while clocked)
"""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, "tools")


def load(d):
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not cc:
        return {}
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        e = rows[r["Dispatch_Id"]]
        e["name"] = r["Kernel_Name"]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            if r["Dispatch_Id"] in rows:
                rows[r["Dispatch_Id"]]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows


def per_kernel(rows, key):
    sel = [e for e in rows.values() if key in e.get("name", "") and e.get("ns")]
    if not sel:
        return None
    big = max(e["ns"] for e in sel)
    sel = [e for e in sel if e["ns"] >= 0.5 * big]
    out = {"calls": len(sel), "ns": sum(e["ns"] for e in sel) / len(sel)}
    for c in sel[0]:
        if c not in ("name", "ns"):
            out[c] = sum(e.get(c, 0.0) for e in sel) / len(sel)
    return out


def main():
    import contextlib
    import io
    argv, sys.argv = sys.argv, [sys.argv[0]]
    with contextlib.redirect_stdout(io.StringIO()):
        import kernel_roofline as kr
    sys.argv = argv
    passes = [load(d) for d in argv[1:]]
    out = []
    for key, (alg, what) in kr.bytes_of.items():
        ps = [per_kernel(p, key) for p in passes]
        ps = [p for p in ps if p]
        if not ps:
            continue
        merged, ns_of = {}, {}
        for p in ps:
            for c, v in p.items():
                if c not in ("calls", "ns") and c not in merged:
                    merged[c] = v; ns_of[c] = p["ns"]
        g = merged.get
        ref_ns = ns_of.get("SQ_WAVES", ps[0]["ns"])
        row = dict(kernel=key, what=what, algorithmic_GB=round(alg / 1e9, 3), large_calls=ps[0]["calls"], ms_under_counters=round(ref_ns / 1e6, 4),
                   algorithmic_TBps_under_counters=round(alg / (ref_ns * 1e-9) / 1e12, 2), counters={c: round(v, 1) for c, v in merged.items()})
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            row["wait_frac"] = round(g("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
            row["issue_frac"] = round(g("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
        vm = g("SQ_INSTS_VMEM_RD", 0.0) + g("SQ_INSTS_VMEM_WR", 0.0)
        if vm and g("SQ_WAVES"):
            row["vmem_per_wave"] = round(vm / g("SQ_WAVES"), 1)
            row["bytes_per_lane"] = round(alg / (64.0 * vm), 2)
            row["valu_per_vmem"] = round(g("SQ_INSTS_VALU", 0.0) / vm, 1)
            if g("SQ_INSTS_SALU") is not None:
                row["salu_per_vmem"] = round(g("SQ_INSTS_SALU") / vm, 1)
        if g("GRBM_GUI_ACTIVE"):
            clk = g("GRBM_GUI_ACTIVE") / 8.0 / ns_of["GRBM_GUI_ACTIVE"]      # shader cycles per ns
            row["effective_clock_ghz"] = round(clk, 3)
            if wc:
                row["waves_per_cu"] = round(4.0 * wc / (clk * ns_of["SQ_WAVE_CYCLES"]) / 256.0, 2)
            if g("VmemLatency") and vm:
                row["vmem_latency_cycles"] = round(g("VmemLatency"), 0)
                row["vmem_in_flight_per_cu"] = round(vm * g("VmemLatency") / (clk * ref_ns) / 256.0, 1)
                row["bytes_in_flight_per_cu"] = round(row["vmem_in_flight_per_cu"] * 64.0 * row["bytes_per_lane"])
        if g("TCP_GATE_EN1"):
            for c in ("TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES"):
                if g(c) is not None:
                    row[c.lower().replace("_cycles", "") + "_frac"] = round(g(c) / g("TCP_GATE_EN1"), 3)
        out.append(row)
    json.dump(dict(what=__doc__.split("\n\n")[0], rows=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
