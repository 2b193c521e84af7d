"""Extract the chemgen fragmentations of octane / STO-3G (BE2, with and without frozen core) that the reference's own
fragmentation tests hold as expected data (tests/_expected_data_for_chemfrag.py, consumed by tests/test_chemfrag.py:160-180)
into tests/golden/fragmentation_chemgen.json.  Data only: the index lists are read as Python literals from the text of that
data file; nothing of the reference is imported or executed.

    python tests/golden/make_golden_chemgen.py          (build container only: needs /root/reference)
"""
import ast
import json
import re
from pathlib import Path

SRC = Path("/root/reference/tests/_expected_data_for_chemfrag.py")
OUT = Path(__file__).resolve().parent / "fragmentation_chemgen.json"
FIELDS = ["AO_per_frag", "AO_per_edge_per_frag", "ref_frag_idx_per_edge_per_frag", "relAO_per_edge_per_frag",
          "relAO_in_ref_per_edge_per_frag", "relAO_per_origin_per_frag", "weight_and_relAO_per_center_per_frag"]
# key of the expected-data dict: (n_BE, basis, iao_valence_basis, frozen_core, wrong_iao_indexing)
CASES = {"chemgen_octane_be2_frozen_core": '(2, "sto-3g", None, True, False): FragPart(',
         "chemgen_octane_be2": '(2, "sto-3g", None, False, False): FragPart(',
         "chemgen_octane_be3_frozen_core": '(3, "sto-3g", None, True, False): FragPart(',
         "chemgen_octane_be3": '(3, "sto-3g", None, False, False): FragPart('}


def literal_after(text, start):
    """The bracketed Python literal that starts at text[start] ('[' or '(')."""
    depth, i = 0, start
    while True:
        c = text[i]
        if c in "[(":
            depth += 1
        elif c in "])":
            depth -= 1
            if depth == 0:
                return ast.literal_eval(text[start:i + 1])
        i += 1


def main():
    text = SRC.read_text()
    out = {}
    for name, key in CASES.items():
        a = text.index(key)
        nxt = re.compile(r"^        \(\d, ", re.M).search(text, a + len(key))
        block = text[a: nxt.start() if nxt else len(text)]
        d = {}
        for f in FIELDS:
            m = re.search(r"^            " + f + r"=", block, re.M)
            d[f] = literal_after(block, m.end())
        d["weight_and_relAO_per_center_per_frag"] = [[float(w), list(c)] for w, c in d["weight_and_relAO_per_center_per_frag"]]
        d["frozen_core"] = bool(re.search(r"^            frozen_core=True", block, re.M))
        out[name] = d
    OUT.write_text(json.dumps(out))
    for k, v in out.items():
        print(k, "fragments", len(v["AO_per_frag"]), "sizes", [len(x) for x in v["AO_per_frag"]], "frozen_core", v["frozen_core"])


if __name__ == "__main__":
    main()
