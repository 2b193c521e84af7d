"""CPU: `bench.py --gpus N` starts N ranks itself (no external launcher), the ranks meet through the library's communicator
(quemb_amd/comm.py; on the hostcheck mock the shared-memory stand-in of the RCCL transport, tests/hostcheck/comm_shm.cpp) or through
torch.distributed gloo, and the JSON line reports what ran.  Also the rendezvous of comm.init_from_env on its own."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "hostcheck"))

SMALL = ["--steps", "1", "--warmup", "0", "--frags-per-gpu", "2", "--n", "24", "--nocc", "4", "--nstreams", "1", "--no-cpu-baseline"]


def _mock():
    import build as hc_build
    return str(hc_build.build())


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "QEMB_RDV_FILE", "QEMB_DIST_BACKEND")}
    env.update(kw)
    return env


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["rccl", "gloo"])
def test_bench_gpus_2_launches_two_ranks(backend):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL, "--lib", _mock()], env=_env(QEMB_DIST_BACKEND=backend),
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2
    assert r["config"]["fragments_per_rank"] == [2, 2]
    assert r["config"]["allreduce_bytes_per_sweep"] > 0
    assert r["scaling"] == "weak" and r["steps"] == 1 and r["warmup"] == 0
    assert r["value"] > 0 and r["fragments_per_s"] > 0
    assert ("library communicator" in r["config"]["transport"]) == (backend == "rccl")


@pytest.mark.timeout(300)
def test_two_ranks_give_the_single_rank_energies():
    """the same 4-fragment ring on 1 rank and on 2: identical sweep energy and residual (the all-reduced buffer is the only coupling)"""
    outs = []
    for gpus, fpg in ((1, 4), (2, 2)):
        args = [a for a in SMALL]
        args[args.index("--frags-per-gpu") + 1] = str(fpg)
        p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), *args, "--lib", _mock()], env=_env(),
                           capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0]))
    a, b = outs
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert abs(a["mean_e_corr_per_sweep"] - b["mean_e_corr_per_sweep"]) < 1e-12
    assert abs(a["residual_norm"] - b["residual_norm"]) < 1e-12
    assert a["ccsd_iterations_per_fragment"] == b["ccsd_iterations_per_fragment"]


def test_gpus_must_match_world_size():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *SMALL, "--lib", _mock()],
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.timeout(120)
def test_failing_rank_fails_the_launcher():
    args = [a for a in SMALL]
    args[args.index("--n") + 1] = "20"        # too small for the synthetic ring: every rank exits with an error
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *args, "--lib", _mock()], env=_env(), capture_output=True, text=True,
                       timeout=100)
    assert p.returncode != 0
    assert "exited with status" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


_RANK_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from quemb_amd import _lib, comm, be_parallel
lib = _lib.declare(C.CDLL(sys.argv[2]))
_lib.check(lib.qemb_init(0), "qemb_init", lib)
rank, world = comm.init_from_env(lib)
assert comm.info(lib) == (rank, world) and be_parallel.world() == (rank, world)
x = np.arange(5, dtype=np.float64) * (rank + 1)
comm.all_reduce(lib, x)
m = comm.all_reduce(lib, np.array([float(rank), -float(rank)]), comm.MAX)
big = np.full(40000, float(rank + 1)); comm.all_reduce(lib, big)          # longer than one pass of the mock transport
buf = np.array([1.0 + rank, 2.0]); be_parallel.all_reduce_sum(buf)
try:
    be_parallel.all_reduce_sum(np.zeros(2), error=ValueError("boom") if rank == 1 else None)
    fail = "none"
except be_parallel.RankFailure as e:
    fail = str(e)
comm.barrier(lib); comm.destroy(lib)
assert comm.info(lib) == (0, 1) and comm.active() is None
print(repr((x.tolist(), m.tolist(), float(big.min()), float(big.max()), buf.tolist(), fail)))
"""


@pytest.mark.timeout(120)
def test_rendezvous_without_a_launcher_file():
    """comm.init_from_env with only RANK / WORLD_SIZE / MASTER_PORT: the ranks of one launch share a parent process, which names the file"""
    world = 3
    procs = [subprocess.Popen([sys.executable, "-c", _RANK_SCRIPT, str(ROOT), _mock()],
                              env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="23456"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=100) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    tot = sum(range(1, world + 1))
    for r, (so, _) in enumerate(outs):
        x, m, bmin, bmax, buf, fail = eval(so.strip().splitlines()[-1])
        assert x == (np.arange(5.0) * tot).tolist() and m == [world - 1.0, 0.0]
        assert bmin == bmax == float(tot)
        assert buf == [float(tot), 2.0 * world]
        assert "1 of 3 rank(s) failed" in fail and (("this rank succeeded" in fail) == (r != 1))


@pytest.mark.timeout(300)
def test_cpu_baseline_runs_whole_fragments_in_both_pool_settings():
    """bench.py's cpu_baseline on tiny sizes (mock device): full solves in the reference's default pool and in the all-cores pool, the
    GPU-side figure on the same fragments, energies agreeing, and the amplitude-update sample at the timed size beside them."""
    args = [a for a in SMALL if a != "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *args, "--full-n", "16", "--full-nocc", "3", "--lib", _mock()],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"].startswith("CCSD iterations/s at n=16") and cb["like_for_like"]["gpu_over_all_cores"] > 0 and cb["value"] > 0 and cb["cores"] >= 4
    fs = cb["full_solve"]
    assert fs["reference_defaults_nproc1_ompnum4"]["nproc"] == 1 and fs["reference_defaults_nproc1_ompnum4"]["ompnum"] == 4
    for key in ("reference_defaults_nproc1_ompnum4", "all_cores"):
        assert fs[key]["iterations_per_s"] > 0 and fs[key]["fragments_per_s"] > 0
        assert fs[key]["max_abs_e_corr_diff_vs_gpu_Eh"] < 1e-8 and fs[key]["max_abs_e_frag_diff_vs_gpu_Eh"] < 1e-8
    assert cb["value"] == fs["all_cores"]["iterations_per_s"]
    assert fs["gpu_same_fragments"]["fragments"] == fs["all_cores"]["nproc"]
    assert cb["n220_amplitude_updates"]["value"] > 0
    assert r["parity_n220_abs_err_Eh"] < 1e-10


@pytest.mark.timeout(600)
def test_world8_strong_scaling_matches_one_rank():
    """`--gpus 8 --scaling strong` with BASELINE configs[2]'s 64 fragments on the mock: eight ranks of eight fragments, the sweep energy,
    residual and iteration counts of the one-rank run of the same 64-fragment ring (world 8 is the driver's node size; it never ran before)."""
    outs = []
    for gpus in (1, 8):
        p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), *SMALL, "--scaling", "strong", "--frags-total", "64",
                            "--lib", _mock()], env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=500)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    a, b = outs
    assert b["n_gpus"] == 8 and b["scaling"] == "strong" and a["scaling"] == "strong"
    assert b["config"]["fragments_per_rank"] == [8] * 8 and a["config"]["fragments_per_rank"] == [64]
    assert abs(a["mean_e_corr_per_sweep"] - b["mean_e_corr_per_sweep"]) < 1e-11
    assert abs(a["residual_norm"] - b["residual_norm"]) < 1e-12
    assert a["ccsd_iterations_per_fragment"] == b["ccsd_iterations_per_fragment"]
    assert b["config"]["allreduce_bytes_per_sweep"] == a["config"]["residual_slots"] * 8 or b["config"]["allreduce_bytes_per_sweep"] > 0
    # weak scaling of the same node: 8 x 2 fragments
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", *SMALL, "--lib", _mock()], env=_env(OMP_NUM_THREADS="1"),
                       capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    w = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert w["scaling"] == "weak" and w["config"]["fragments_per_rank"] == [2] * 8


_DYING_RANK_SCRIPT = r"""
import ctypes as C, os, signal, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from quemb_amd import _lib, comm
lib = _lib.declare(C.CDLL(sys.argv[2]))
_lib.check(lib.qemb_init(0), "qemb_init", lib)
rank, world = comm.init_from_env(lib)
comm.all_reduce(lib, np.ones(3))                 # everybody is there
if rank == int(sys.argv[3]):
    os.kill(os.getpid(), signal.SIGKILL)         # dies like an OOM-killed process: no exception, no goodbye
t0 = time.monotonic()
try:
    comm.all_reduce(lib, np.ones(3))
except _lib.QembError as e:
    print(f"rank {rank}: {e} after {time.monotonic() - t0:.1f} s", flush=True)
    try:
        comm.all_reduce(lib, np.ones(3))          # and it stays failed, at once
    except _lib.QembError:
        pass
    assert time.monotonic() - t0 < float(os.environ["QEMB_COMM_TIMEOUT_S"]) + 10.0
    sys.exit(3)
sys.exit(0)                                        # must not happen: a rank was gone
"""


@pytest.mark.timeout(180)
def test_killed_rank_ends_the_other_seven_within_the_bound():
    """world 8 under an EXTERNAL launcher (this test): rank 5 is killed between two collectives; the other seven must come back from the
    all-reduce with an error inside QEMB_COMM_TIMEOUT_S and exit non-zero -- nobody hangs, nobody is re-executed."""
    import time
    world, victim, bound = 8, 5, 6.0
    t0 = time.monotonic()
    procs = [subprocess.Popen([sys.executable, "-c", _DYING_RANK_SCRIPT, str(ROOT), _mock(), str(victim)],
                              env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="23471",
                                       QEMB_COMM_TIMEOUT_S=str(bound)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=150) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        if r == victim:
            assert p.returncode == -9
        else:
            assert p.returncode == 3, (r, p.returncode, se[-1500:])
            assert "a rank is gone" in so
    assert time.monotonic() - t0 < 120


_C5_SCRIPT = r"""
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1])
from quemb_amd import _lib
lib = _lib.declare(C.CDLL(sys.argv[2]))
_lib.check(lib.qemb_init(0), "qemb_init", lib)
import bench
r = bench.kbe_c5_sweeps(lib, reps=2)
r["oracle_imported"] = any(m.startswith("qemb_oracle") for m in sys.modules)
print("RESULT " + json.dumps(r))
"""


@pytest.mark.timeout(600)
def test_kbe_c5_bench_field_on_the_mock():
    """bench.py's `kbe_c5` section (BASELINE configs[4] at its own dimensions) on the mock: the model's mean field comes from the device
    RHF -- the oracle is not imported on this path --, four fragments of 36 embedding orbitals, HF-in-HF, sweep statistics."""
    p = subprocess.run([sys.executable, "-c", _C5_SCRIPT, str(ROOT), _mock()], env=_env(), capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert r["oracle_imported"] is False
    assert r["fragments"] == 4 and r["n_emb"] == [36] * 4 and r["aos_per_cell"] == 24 and r["kpts"] == 3 and r["electrons_per_cell"] == 28
    assert abs(r["hf_in_hf_error_Eh"]) < 1e-8 and r["e_corr_per_cell"] < -1e-3
    assert r["sweep"]["sweeps"] == 2 and r["sweep"]["p50_ms"] <= r["sweep"]["max_ms"] and r["sweep_ms"] > 0


@pytest.mark.timeout(300)
def test_comm_startup_failure_falls_back_to_gloo():
    """If the library communicator does not come up (it never ran with more than one rank on real GPUs), the N-rank bench must still deliver
    its line: the ranks fall back to a torch.distributed gloo group for the one small all-reduce per sweep and say so in config.transport."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL, "--lib", _mock()], env=_env(QEMB_BENCH_FAIL_COMM_INIT="1"),
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["config"]["fragments_per_rank"] == [2, 2] and r["value"] > 0
    assert "FALLBACK" in r["config"]["transport"] and "falling back to torch.distributed gloo" in p.stderr
    # same sweep as the two-rank run over the library communicator
    q = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL, "--lib", _mock()], env=_env(), capture_output=True, text=True, timeout=280)
    r0 = json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][0])
    assert abs(r["mean_e_corr_per_sweep"] - r0["mean_e_corr_per_sweep"]) < 1e-12 and abs(r["residual_norm"] - r0["residual_norm"]) < 1e-12
