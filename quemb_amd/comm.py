"""comm -- the library's own communicator (RCCL behind the C ABI: qemb_comm_*, include/qemb_hip.h), one process per GPU.

The reference has no communication backend (molbe/be_parallel.py:484-517 returns pickled result tuples through pathos pipes); the
sharded sweep here needs exactly one all-reduce per sweep (be_parallel.be_func_parallel).  This module only carries the 128-byte
communicator id from rank 0 to the other ranks -- plain files, no Python package beyond the standard library -- and wraps the calls.

    comm.init_from_env(lib)            # RANK / WORLD_SIZE (torchrun's variables, or bench.py's own launcher); world 1: no-op
    comm.all_reduce(lib, buf)          # in place on a float64 numpy buffer, every rank gets the identical result
    comm.barrier(lib); comm.destroy(lib)

Rendezvous: rank 0 writes the id to a file every rank can name and renames it into place; the others poll for it.  The name is
QEMB_RDV_FILE when the launcher provides one (bench.py does), otherwise it is derived from what the ranks of ONE launch on ONE node
have in common and no other launch has: the parent process (pid + start time) and MASTER_PORT.  Multi-node launchers pass the id
themselves: `init(lib, rank, world, id_bytes)` after broadcasting `unique_id(lib)` their own way (MPI, a torch store, ...).
"""

from __future__ import annotations

import ctypes as C
import os
import tempfile
import time

import numpy as np

from ._lib import check

ID_BYTES = 128
SUM, MAX = 0, 1
_active = None          # the library whose communicator this process created (be_parallel picks it up from here)


def active():
    """(lib, rank, world) of the communicator created by init / init_from_env in this process, or None."""
    if _active is None:
        return None
    r, w = info(_active)
    return (_active, r, w) if w > 1 or _force_single else None


_force_single = False   # tests: treat a one-rank communicator as active so that the collective itself runs


def unique_id(lib) -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    check(lib.qemb_comm_unique_id(buf), "qemb_comm_unique_id", lib)
    return buf.raw


def init(lib, rank: int, world: int, id_bytes: bytes):
    if len(id_bytes) != ID_BYTES:
        raise ValueError(f"communicator id must be {ID_BYTES} bytes")
    global _active
    check(lib.qemb_comm_init(int(rank), int(world), C.create_string_buffer(id_bytes, ID_BYTES)), "qemb_comm_init", lib)
    _active = lib


def info(lib):
    r, w = C.c_int(), C.c_int()
    check(lib.qemb_comm_info(C.byref(r), C.byref(w)), "qemb_comm_info", lib)
    return r.value, w.value


def all_reduce(lib, buf: np.ndarray, op: int = SUM) -> np.ndarray:
    if buf.dtype != np.float64 or not buf.flags.c_contiguous:
        raise ValueError("all_reduce works in place on a C-contiguous float64 buffer")
    check(lib.qemb_comm_allreduce(buf.ctypes.data, buf.size, int(op)), "qemb_comm_allreduce", lib)
    return buf


def barrier(lib):
    all_reduce(lib, np.zeros(1))


def destroy(lib):
    global _active
    check(lib.qemb_comm_destroy(), "qemb_comm_destroy", lib)
    if _active is lib:
        _active = None


def _parent_start_ticks() -> str:
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]      # field 22 of proc(5): starttime
    except Exception:  # noqa: BLE001
        return "0"


def rendezvous_file() -> str:
    p = os.environ.get("QEMB_RDV_FILE")
    if p:
        return p
    uid = os.getuid() if hasattr(os, "getuid") else 0
    return os.path.join(tempfile.gettempdir(),
                        f"qemb_rdv_{uid}_{os.getppid()}_{_parent_start_ticks()}_{os.environ.get('MASTER_PORT', '0')}")


def init_from_env(lib, timeout_s: float = 600.0):
    """Create the communicator from RANK / WORLD_SIZE.  Returns (rank, world).  qemb_init(device) must have been called."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    path = rendezvous_file()
    if rank == 0:
        uid = unique_id(lib)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)          # appears atomically with its full contents
    else:
        t0 = time.monotonic()
        while True:
            try:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == ID_BYTES:
                    break
            except FileNotFoundError:
                pass
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout_s:.0f} s (did rank 0 start?)")
            time.sleep(0.02)
    init(lib, rank, world, uid)         # collective: returns when every rank has joined
    barrier(lib)
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return rank, world
