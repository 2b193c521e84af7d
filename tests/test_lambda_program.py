"""CPU: the generated contraction program of the device Lambda solver (tools/gen_cc_lambda.py), interpreted with NumPy,
against the oracle's reverse-mode tape (oracle/qemb_oracle/ccsd_lambda.py): amplitude numerators, t cotangents (the Lambda
iteration), Fock cotangents (1-RDM) and ERI-block cotangents (2-RDM)."""
import importlib.util
import sys

import numpy as np

from helpers import ROOT, synthetic_fragment
from qemb_oracle import ccsd, ccsd_lambda, scf

spec = importlib.util.spec_from_file_location("gen_cc_lambda", ROOT / "tools" / "gen_cc_lambda.py")
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def run_program(t1, t2, eris, z1, z2):
    o, v = t1.shape
    dim = {"o": o, "v": v}
    fwd, bwd, zero, tdep = gen.analyse()
    names = gen.all_tensors(bwd)
    val = {n: np.zeros([dim[c] for c in sig]) for n, (sig, kind) in names.items() if kind != "virtual"}
    val["t1"], val["t2"] = t1, t2
    for b in ("oooo", "ovoo", "ovov", "oovv", "ovvo", "ovvv"):
        val[b] = getattr(eris, b)
    Vl = eris.vvvv.transpose(0, 2, 1, 3)          # Vl[a,b,c,d] = (ac|bd)

    def exe(s):
        if s[0] == "es":
            val[s[1]] += s[2] * np.einsum(s[3], val[s[4]], val[s[5]], optimize=True)
        elif s[0] == "perm":
            val[s[1]] += s[2] * np.einsum(s[3], val[s[4]])
        else:
            val[s[1]] += s[2] * np.einsum("abcd,ijcd->ijab", Vl, val[s[3]])
    for st, skip in fwd:
        if not skip:
            exe(st)
    val["n1_bar"] += z1; val["n2_bar"] += z2
    for s in bwd:
        exe(s[:-1])
    return val, bwd


def test_generated_program_matches_the_oracle_tape():
    n, o = 7, 3
    h, e1 = synthetic_fragment(n, o, 5, scale=0.12)
    mf = scf.rhf(h, e1, o)
    eris = ccsd.Eris(e1, mf["mo_coeff"], o, mo_energy=mf["mo_energy"])
    conv, e, t1, t2, _ = ccsd.kernel(eris, conv_tol=1e-13, conv_tol_normt=1e-11)
    rng = np.random.default_rng(3)
    z1 = 0.1 * rng.standard_normal(t1.shape)
    z2 = 0.1 * rng.standard_normal(t2.shape); z2 = z2 + z2.transpose(1, 0, 3, 2)
    val, bwd = run_program(t1, t2, eris, z1, z2)
    n1, n2 = ccsd.amplitude_numerators(t1, t2, eris)
    assert np.abs(val["n1"] - n1).max() < 1e-13 and np.abs(val["n2"] - n2).max() < 1e-13
    lag = ccsd_lambda.Lagrangian(t1, t2, eris)
    g = lag.vjp(z1, z2)
    assert np.abs(val["t1_bar"] - (g["t1"] + lag.eia * z1)).max() < 1e-12
    # t2[i,j,a,b] = t2[j,i,b,a]: equivalent rewritings of the equations differ in the part of the t2 cotangent that is
    # antisymmetric under that swap, which no symmetric variation can see; the symmetric part is the gradient
    sym = lambda x: 0.5 * (x + x.transpose(1, 0, 3, 2))
    assert np.abs(sym(val["t2_bar"]) - sym(g["t2"] + lag.eijab * z2)).max() < 1e-12
    fbar = np.zeros((n, n))
    fbar[:o, :o] = val["dfoo_bar"]; fbar[:o, o:] = val["fov_bar"]; fbar[o:, o:] = val["dfvv_bar"]
    assert np.abs(fbar - g["fock"]).max() < 1e-12
    for b in ("oooo", "ovoo", "ovov", "oovv", "ovvo", "ovvv"):
        assert np.abs(val[b + "_bar"] - g[b]).max() < 1e-12, b
    assert np.abs(val["vvvv_l_bar"].transpose(0, 2, 1, 3) - g["vvvv"]).max() < 1e-12
    # the Lambda-iteration subset alone reproduces the t cotangents
    val2 = {k: (np.zeros_like(x) if k.endswith("_bar") else x) for k, x in val.items()}
    val2["n1_bar"] = z1.copy(); val2["n2_bar"] = z2.copy()
    Vl = eris.vvvv.transpose(0, 2, 1, 3)
    for s in bwd:
        if not s[-1]["lam"]:
            continue
        if s[0] == "es":
            val2[s[1]] += s[2] * np.einsum(s[3], val2[s[4]], val2[s[5]], optimize=True)
        elif s[0] == "perm":
            val2[s[1]] += s[2] * np.einsum(s[3], val2[s[4]])
        else:
            val2[s[1]] += s[2] * np.einsum("abcd,ijcd->ijab", Vl, val2[s[3]])
    assert np.abs(val2["t1_bar"] - val["t1_bar"]).max() < 1e-13 and np.abs(val2["t2_bar"] - val["t2_bar"]).max() < 1e-13



def test_committed_program_is_the_generator_output(tmp_path):
    """quemb_amd/csrc/cc_lambda_program.inc is generated; it must be exactly what tools/gen_cc_lambda.py emits."""
    out = tmp_path / "prog.inc"
    gen.emit(str(out))
    assert out.read_text() == (ROOT / "quemb_amd" / "csrc" / "cc_lambda_program.inc").read_text()
