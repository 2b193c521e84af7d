"""The three statements of the options struct -- include/qemb_hip.h, quemb_amd/_lib.py and the binding INTEGRATION.md tells a QuEmb
maintainer to write -- must agree field for field, and the library must reject a struct that was not initialised against this header."""
import ctypes as C
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "hostcheck"))

CTYPE = {"double": "c_double", "int": "c_int", "uint32_t": "c_uint32"}


def header_fields():
    txt = (ROOT / "include" / "qemb_hip.h").read_text()
    body = txt[txt.index("typedef struct {"):txt.index("} qemb_solver_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [(m.group(2), CTYPE[m.group(1)]) for m in re.finditer(r"\b(double|int|uint32_t)\s+(\w+)\s*;", body)]


def documented_fields():
    txt = (ROOT / "INTEGRATION.md").read_text()
    blk = txt[txt.index("class SolverOpts(C.Structure)"):]
    blk = blk[:blk.index("]\n") + 1]
    return [(m.group(1), m.group(2)) for m in re.finditer(r'\("(\w+)",\s*C\.(\w+)\)', blk)]


def test_options_struct_is_stated_identically_three_times():
    from quemb_amd._lib import SolverOpts
    mirror = [(n, t.__name__) for n, t in SolverOpts._fields_]
    # c_uint32 is an alias of c_uint on this platform: compare by size and signedness through the ctypes objects themselves
    canon = lambda fields: [(n, C.sizeof(getattr(C, t)), getattr(C, t)(-1).value < 0 if "double" not in t else None) for n, t in fields]
    assert canon(header_fields()) == canon(mirror) == canon(documented_fields())
    assert header_fields()[0][0] == "struct_size" and header_fields()[-1][0] == "strict_convergence"


def test_documented_check_handles_the_warning_status():
    txt = (ROOT / "INTEGRATION.md").read_text()
    assert "if rc < 0: raise" in txt and "if rc > 0: warnings.warn" in txt
    assert "lib.qemb_default_opts(C.byref(opts))" in txt


def test_struct_size_mismatch_is_rejected():
    import build as hc_build
    from quemb_amd import _lib
    from quemb_amd._lib import SolverOpts
    lib = _lib.declare(C.CDLL(str(hc_build.build())))
    good = SolverOpts(); lib.qemb_default_opts(C.byref(good))
    assert good.struct_size == C.sizeof(SolverOpts) and good.strict_convergence == 1
    n, o = 6, 2
    fr = C.c_void_p(); _lib.check(lib.qemb_frag_create(n, 2, C.byref(fr)), lib=lib)
    s4 = np.zeros((n * (n + 1) // 2,) * 2); _lib.check(lib.qemb_frag_set_eri_s4(fr, s4.ctypes.data), lib=lib)
    h = np.diag(np.arange(n, dtype=float))
    mo, eps = np.empty((n, n)), np.empty(n)
    es, cv, cy = C.c_double(), C.c_int(), C.c_int()
    call = lambda op: lib.qemb_frag_scf(fr, o, h.ctypes.data, None, C.byref(op) if op is not None else None, mo.ctypes.data, eps.ctypes.data,
                                        None, None, C.byref(es), C.byref(cv), C.byref(cy))
    assert call(good) == 0 and call(None) == 0            # NULL = defaults
    for bad_size in (0, C.sizeof(SolverOpts) - 8, C.sizeof(SolverOpts) + 8):
        bad = SolverOpts.from_buffer_copy(good); bad.struct_size = bad_size
        assert call(bad) == -1
        assert "struct_size" in lib.qemb_last_error().decode()
    raw = SolverOpts()                                     # never initialised: zeros
    assert call(raw) == -1
    lib.qemb_frag_free(fr)


def test_design_figures_follow_the_tracked_profile():
    """DESIGN.md section 5 quotes ladder / ring figures 'from profiles/rNN_bench_nstreams1_kernel_stats.csv': the block between the figures
    markers must be exactly what tools/design_figures.py derives from that tracked file (round-3 review: text and artifact had drifted)."""
    sys.path.insert(0, str(ROOT / "tools"))
    import design_figures
    txt = (ROOT / "DESIGN.md").read_text()
    m = re.search(r"<!-- figures: (\S+) -->\n```\n(.*?)\n```\n<!-- /figures -->", txt, flags=re.S)
    assert m, "DESIGN.md carries no figures block"
    src = m.group(1)
    assert (ROOT / src).exists(), f"{src} is not tracked"
    import os
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        want = design_figures.figures(src)
    finally:
        os.chdir(cwd)
    assert m.group(2).strip() == want.strip()
    # and the prose of the section quotes the same averages
    lad = re.search(r"ladder \(\+\) pairs: \d+ dispatches, ([0-9.]+) ms", want).group(1)
    assert lad in txt.replace(m.group(0), ""), f"the text of DESIGN.md does not quote the ladder (+) average {lad} ms of {src}"


def test_no_tracked_roofline_row_exceeds_its_peak():
    """A kernel cannot run above the roofline it is priced against: a row with frac > 1 in a tracked per-kernel roofline table is a mis-attributed
    flop or byte count (round 4: a factor-route product priced with the four-index route's flops showed 20.7 x the FP64 matrix peak)."""
    import json
    files = sorted((ROOT / "profiles").glob("*kernel_roofline*.jsonl"))
    assert files, "no per-kernel roofline table is tracked"
    for f in files:
        for ln in f.read_text().splitlines():
            if not ln.strip().startswith("{"):
                continue
            d = json.loads(ln)
            for k, val in d.items():
                if k.startswith("frac_of") and isinstance(val, (int, float)):
                    assert val <= 1.0, f"{f.name}: {d.get('kernel')} at {val} of its peak"


def test_readme_ladder_agreement_lines_follow_their_csv():
    """profiles/README.md quotes, per round, the rocprofv3 averages of the two ladder dispatches 'X + Y ms ... (`rNN_bench_nstreams1_kernel_stats.csv`)': the
    quoted figures must be the ones in that tracked file (round 4 shipped 2.770 + 2.364 beside a CSV that says 2.823 + 2.409)."""
    import csv
    txt = (ROOT / "profiles" / "README.md").read_text()
    hits = re.findall(r"rocprofv3\s+([0-9.]+) \+ ([0-9.]+) ms = [0-9.]+ ms per dispatch \(`(r\d\d_bench_nstreams1_kernel_stats.csv)`", txt)
    assert hits, "no ladder agreement line with its CSV in profiles/README.md"
    for a, b, name in hits:
        rows = {r["Name"]: r for r in csv.DictReader(open(ROOT / "profiles" / name))}
        def avg(frag):
            hit = [r for nm, r in rows.items() if "dgemm_mfma_kernel<" + frag + ">" in nm]
            assert len(hit) == 1
            return float(hit[0]["AverageNs"]) * 1e-6
        assert abs(avg("7, 2, 2, 4, 16, true, true, 2, 1, 1") - float(a)) < 5e-4 and abs(avg("6, 2, 2, 4, 16, true, true, 2, 1, 1") - float(b)) < 5e-4, (name, a, b)
