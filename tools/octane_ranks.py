"""Small-fragment regime: octane BE2 / BE3 sweeps with R ranks sharing ONE GPU (gloo all-reduce), to show how kernel-latency-bound
fragments overlap across processes.   python -m torch.distributed.run --nproc-per-node R tools/octane_ranks.py
(keep R <= 5 on a gpurun box: its process guard allows 6 processes on the GPU)"""
import os, sys, time
sys.path.insert(0, ".")
from pathlib import Path
import torch
import torch.distributed as dist
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group(backend="gloo")
from quemb_amd import _lib
_lib.init(0)
from quemb_amd.fragpart import FragPart
from quemb_amd.integrals import RHF, Mole
from quemb_amd.mbe import BE
G = Path("tests/golden")
mf = RHF(Mole(G / "octane.xyz")); mf.kernel()
for key in ("test_autogen_octane_be2", "test_autogen_octane_be3"):
    be = BE(mf, FragPart.from_json(G / "fragmentation.json", key))
    be.oneshot()
    if world > 1: dist.barrier()
    t = time.time()
    for _ in range(5):
        e, _ = be.oneshot()
    if world > 1: dist.barrier()
    dt = (time.time() - t) / 5
    if be.rank == 0:
        print("RESULT %s ranks=%d sweep %.1f ms E_corr %.10f" % (key, world, dt * 1e3, e), flush=True)
if world > 1:
    dist.destroy_process_group()
