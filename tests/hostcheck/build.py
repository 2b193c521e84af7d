"""Build tests/hostcheck/libqemb_hostcheck.so: the product's DRIVER sources linked against the scalar mock
device layer (dev_ops_cpu.cpp).  Test infrastructure only -- see the header of dev_ops_cpu.cpp."""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent.parent / "quemb_amd" / "csrc"
OUT = HERE / "libqemb_hostcheck.so"


def build(force=False):
    srcs = sorted(CSRC.glob("*.cpp")) + [HERE / "dev_ops_cpu.cpp"]
    deps = srcs + sorted(CSRC.glob("*.h")) + [CSRC.parent.parent / "include" / "qemb_hip.h"]
    if OUT.exists() and not force and all(OUT.stat().st_mtime > d.stat().st_mtime for d in deps):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wl,--no-undefined", "-DQEMB_HOSTCHECK", f"-I{CSRC}", "-o", str(OUT)] + [str(s) for s in srcs]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
