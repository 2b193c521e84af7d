"""qemb_oracle -- NumPy restatement of QuEmb's per-fragment hot path.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from quemb_amd.
"""
from . import be, ccsd, eri, rdm, schmidt, scf  # noqa: F401
