"""quemb_amd -- MI355X (gfx950) per-fragment embedding solver behind QuEmb's Frags / solver seams.

Only the hot path of troyvvgroup/quemb is here: Schmidt decomposition, AO->fragment
ERI transforms (dense + DF), fragment RHF, per-fragment CCSD, fragment energies and the density-matching
residual.  All numerics run in hand-written HIP kernels inside ``libqemb_hip.so`` (C ABI in
``include/qemb_hip.h``); this package is the thin Python host side mirroring the reference's names.
There is no CPU fallback: importing is cheap, but the first call that needs the device raises if the
library or a GPU is missing.
"""

from .hostthreads import limit_blas_threads as _limit_blas_threads

_limit_blas_threads()      # NumPy's BLAS pool inside the CPU share of the process (hostthreads.py: the cgroup-throttling stalls)

__all__ = ["_lib"]
