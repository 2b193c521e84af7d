"""Build tests/hostcheck/libqemb_hostcheck.so: the product's DRIVER sources linked against the scalar mock
device layer (dev_ops_cpu.cpp).  Test infrastructure only -- see the header of dev_ops_cpu.cpp."""
import fcntl
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent.parent / "quemb_amd" / "csrc"
OUT = HERE / "libqemb_hostcheck.so"


def build(force=False):
    # QEMB_HOSTCHECK_LIB: use a prebuilt variant instead (e.g. an -fsanitize=address,undefined build, see tests/hostcheck/README)
    if os.environ.get("QEMB_HOSTCHECK_LIB"):
        return Path(os.environ["QEMB_HOSTCHECK_LIB"])
    srcs = sorted(CSRC.glob("*.cpp")) + [HERE / "dev_ops_cpu.cpp", HERE / "comm_shm.cpp"]
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + [CSRC.parent.parent / "include" / "qemb_hip.h", CSRC.parent.parent / "include" / "qemb_hip_ops.h"]
    def fresh():
        return OUT.exists() and all(OUT.stat().st_mtime > d.stat().st_mtime for d in deps)
    if fresh() and not force:
        return OUT
    # several test processes (the gloo ranks) may get here together: one builds, the others wait; the library appears atomically
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if fresh() and not force:
            return OUT
        tmp = OUT.with_suffix(f".tmp{os.getpid()}.so")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wl,--no-undefined", "-DQEMB_HOSTCHECK", f"-I{CSRC}", "-o", str(tmp)] + [str(s) for s in srcs] + ["-lrt"]
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, OUT)
        finally:
            if tmp.exists():
                tmp.unlink()
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
