"""Shared test helpers (paths, synthetic fragment family of SURVEY.md section 8(d))."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT / "oracle") not in sys.path:
    sys.path.insert(0, str(ROOT / "oracle"))
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def load_frag_lists(key):
    return json.loads((GOLDEN / "fragmentation.json").read_text())[key]


def synthetic_scale(n):
    """ERI amplitude of the synthetic family.  SURVEY.md 8(d) fixes 0.06; with naux = 3n the Coulomb matrix elements grow
    like n, and the oracle shows the fragment RHF/CCSD diverging for n >~ 100 at that value (HF gap -> 0.1), so the
    amplitude is held at 0.06 for n <= 55 and scaled as n^-1/2 beyond (0.03 at n = 220)."""
    return 0.06 * min(1.0, (55.0 / n) ** 0.5)


def synthetic_fragment(n, o, seed, naux=None, scale=None, gap=2.0):
    """SURVEY.md 8(d) synthetic family: DF-factorised 8-fold-symmetric PSD ERIs + gapped one-body part.
    Returns (h, eri_s1)."""
    rng = np.random.default_rng(seed)
    naux = naux or 3 * n
    scale = synthetic_scale(n) if scale is None else scale
    B = scale * rng.standard_normal((naux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    eri = np.einsum("Ppq,Prs->pqrs", B, B, optimize=True)
    A = rng.standard_normal((n, n))
    h = np.diag(gap * np.arange(n)) + 0.3 * 0.5 * (A + A.T)
    return h, eri
