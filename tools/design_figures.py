"""The ladder / ring figures DESIGN.md section 5 quotes, derived from a tracked rocprofv3 --kernel-trace --stats summary.

    python tools/design_figures.py profiles/r04_bench_nstreams1_kernel_stats.csv

Prints the block that DESIGN.md carries between `<!-- figures: ... -->` and `<!-- /figures -->`; tests/test_boundary_docs.py re-derives
it from the same file, so the text and the artifact cannot drift apart (round-3 review, weak 9)."""
import csv
import sys

O, V = 20, 200
PEAK = 78.6


def figures(path):
    npo, nmo, npv, nmv, nov = O * (O + 1) // 2, O * (O - 1) // 2, V * (V + 1) // 2, V * (V - 1) // 2, O * V
    rows = {r["Name"]: r for r in csv.DictReader(open(path))}

    def pick(frag):
        hit = [r for nm, r in rows.items() if "dgemm_mfma_kernel<" + frag + ">" in nm]
        if len(hit) != 1:
            raise SystemExit(f"{path}: {len(hit)} kernels match {frag}")
        return int(hit[0]["Calls"]), float(hit[0]["AverageNs"]) * 1e-6

    cp, tp = pick("7, 2, 2, 4, 16, true, true, 2, 1, 1")
    cm, tm = pick("6, 2, 2, 4, 16, true, true, 2, 1, 1")
    cr, tr = pick("4, 4, 2, 4, 16, true, true, 2, 0, 1")
    f_lad = 2.0 * npo * npv * npv + 2.0 * nmo * nmv * nmv
    f_ring = 2.0 * float(nov) ** 3
    tf_lad = f_lad / ((tp + tm) * 1e-3) / 1e12
    tf_ring = f_ring / (tr * 1e-3) / 1e12
    return "\n".join([
        f"source: {path}",
        f"ladder (+) pairs: {cp} dispatches, {tp:.3f} ms each",
        f"ladder (-) pairs: {cm} dispatches, {tm:.3f} ms each",
        f"ladder pair: {tp + tm:.3f} ms, {f_lad / 1e9:.1f} GFLOP executed = {tf_lad:.1f} TFLOP/s = {tf_lad / PEAK:.3f} of {PEAK}",
        f"ring product: {cr} dispatches, {tr:.3f} ms each, {f_ring / 1e9:.1f} GFLOP = {tf_ring:.1f} TFLOP/s = {tf_ring / PEAK:.3f} of {PEAK}",
    ])


if __name__ == "__main__":
    print(figures(sys.argv[1]))
