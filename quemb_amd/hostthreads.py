"""Host threads of the Python side: keep NumPy's BLAS pool inside the CPU share this process really has.

Measured (round 4, tools/octane_idle_probe.py, profiles/r04_octane_idle_probe.log): on a GPU box whose container is limited to 16 CPUs
(cgroup `cpu.max` = 1600000 / 100000) but shows 256 cores, OpenBLAS starts one worker per visible core and the workers spin for a while
after every NumPy call.  The host work around the device calls -- building a BE object, update_heff, the residual -- burns the cgroup's
CPU quota of the 100 ms period in a few ms, the kernel throttles the WHOLE process until the period ends, and with it the threads that
feed the GPU: a sweep that takes 15 ms then takes 90-100 (the "92 ms sweep" of the round-3 octane figures; `nr_throttled` in
/sys/fs/cgroup/cpu.stat counts them).  With the BLAS pool capped at the usable cores the stalls are gone.

The reference sets its thread counts itself (`OMP_NUM_THREADS`, molbe/be_parallel.py:476, solver.py:992); this module does the same for
the one pool that matters here.  `limit_blas_threads()` runs at package import; QEMB_KEEP_BLAS_THREADS=1 leaves everything alone."""
from __future__ import annotations

import os


def usable_cores() -> int:
    """cores this process may use: scheduler affinity, capped by the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us`)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


_limiter = None


def limit_blas_threads(max_threads: int | None = None) -> int | None:
    """Cap the BLAS / OpenMP pools of the process at `max_threads` (default: half of the usable cores, at least 1 -- the other half is for
    the threads that drive the device: one per fragment in flight, plus the runtime's own).  Pools that are already smaller stay as they
    are.  Returns the cap, or None when nothing was done."""
    global _limiter
    if os.environ.get("QEMB_KEEP_BLAS_THREADS", "0") not in ("", "0"):
        return None
    # several ranks on one node (one process per GPU) share the node's CPU quota: LOCAL_WORLD_SIZE (torchrun, bench.py's launcher) divides it
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    except ValueError:
        local_world = 1
    cap = int(max_threads) if max_threads else max(1, usable_cores() // (2 * local_world))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # pools that have not started yet read these
        try:
            if int(os.environ.get(var, "0") or 0) > cap or var not in os.environ:
                os.environ[var] = str(cap)
        except ValueError:
            os.environ[var] = str(cap)
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        if any(int(p.get("num_threads", 1)) > cap for p in info):
            _limiter = threadpoolctl.threadpool_limits(limits=cap)      # kept alive: the limit lasts as long as the object
    except Exception:  # noqa: BLE001  (no threadpoolctl: the environment variables above cover pools created later)
        pass
    return cap
