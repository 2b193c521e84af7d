"""Generator of the contraction program behind the CCSD Lambda equations / response densities on the device.

The RCCSD amplitude equations that solve_ccsd runs (molbe/solver.py:900-907; PySCF cc/rccsd.py update_amps, restated in
SURVEY.md Appendix A) are written below ONCE as a list of binary contractions (`FORWARD`).  Everything relax_density=True
needs (solver.py:925-939: Lambda amplitudes, `make_rdm1`, `make_rdm2(with_dm1=False)`) is a derivative of the Lagrangian
L = E + z.r, so the reverse sweep of that list -- generated here, statement by statement -- is the whole algorithm:

    Lambda iteration   : t_bar = dE/dt + (dn/dt)^T z           (statements that reach t1_bar / t2_bar)
    response densities : f_bar = dL/df,  V_bar(block) = dL/dV   (all statements, run once at convergence)

`python tools/gen_cc_lambda.py` writes quemb_amd/csrc/cc_lambda_program.inc (tensor table + forward + backward statement
tables executed by cc_lambda.cpp through dev_gemm / dev_copy4 / the (+/-) ladder).  tests/test_lambda_program.py interprets
the same tables with NumPy and checks them against the oracle.

Index letters: i j k l m n = occupied, a b c d e f = virtual.  Statement kinds:
    ("es",   dst, coef, "A,B->D", a, b)   dst[D] += coef * sum a[A] b[B]
    ("perm", dst, coef, "A->D",   a)      dst[D] += coef * a[A]
    ("ladder", dst, coef, a)              dst[ijab] += coef * sum_cd (ac|bd) a[ijcd]     (a symmetric under ij<->ji, cd<->dc)
"""
import sys
from pathlib import Path

OCC, VIR = "ijklmn", "abcdef"


def space(idx):
    return "".join("o" if c in OCC else "v" for c in idx)


# name -> (space signature, kind).  kinds: t (amplitudes), int (integral block), zero (Fock input whose value is zero for
# eris.fock = diag(mo_energy), solver.py:901-902, but whose cotangent is the 1-particle density), fwd (intermediate)
TENSORS = {}


def T(name, sig, kind="fwd"):
    TENSORS[name] = (sig, kind)
    return name


for _n, _s in (("t1", "ov"), ("t2", "oovv")):
    T(_n, _s, "t")
for _n, _s in (("oooo", "oooo"), ("ovoo", "ovoo"), ("ovov", "ovov"), ("oovv", "oovv"), ("ovvo", "ovvo"), ("ovvv", "ovvv")):
    T(_n, _s, "int")
for _n, _s in (("fov", "ov"), ("dfoo", "oo"), ("dfvv", "vv")):
    T(_n, _s, "zero")

FORWARD = []


def es(dst, coef, subs, a, b):
    ins, out = subs.split("->")
    ia, ib = ins.split(",")
    assert space(ia) == TENSORS[a][0] and space(ib) == TENSORS[b][0], (subs, a, b)
    if dst not in TENSORS:
        T(dst, space(out))
    assert space(out) == TENSORS[dst][0], (subs, dst)
    FORWARD.append(("es", dst, float(coef), subs, a, b))


def perm(dst, coef, subs, a):
    ia, out = subs.split("->")
    assert space(ia) == TENSORS[a][0], (subs, a)
    if dst not in TENSORS:
        T(dst, space(out))
    assert space(out) == TENSORS[dst][0] and sorted(ia) == sorted(out), (subs, dst)
    FORWARD.append(("perm", dst, float(coef), subs, a))


def ladder(dst, coef, a):
    FORWARD.append(("ladder", dst, float(coef), a))


# ------------------------------------------------------------------------------------------------------------------
# The amplitude numerators (oracle/qemb_oracle/ccsd.py amplitude_numerators, ternary products split), outputs n1, n2, E
# ------------------------------------------------------------------------------------------------------------------
perm("tau", 1, "ijab->ijab", "t2"); es("tau", 1, "ia,jb->ijab", "t1", "t1")
perm("Lovov", 2, "kcld->kcld", "ovov"); perm("Lovov", -1, "kdlc->kcld", "ovov")
perm("Lovoo", 2, "lcki->lcki", "ovoo"); perm("Lovoo", -1, "kcli->lcki", "ovoo")
perm("Lovvv", 2, "kdac->kdac", "ovvv"); perm("Lovvv", -1, "kcad->kdac", "ovvv")
es("Foo", 1, "kcld,ilcd->ki", "Lovov", "tau"); perm("Foo", 1, "ki->ki", "dfoo")
es("Fvv", -1, "kcld,klad->ac", "Lovov", "tau"); perm("Fvv", 1, "ac->ac", "dfvv")
es("Fov", 1, "kcld,ld->kc", "Lovov", "t1"); perm("Fov", 1, "kc->kc", "fov")
es("FT", 1, "kc,ic->ki", "fov", "t1")
es("FTv", 1, "kc,ka->ac", "fov", "t1")
perm("Loo", 1, "ki->ki", "Foo"); perm("Loo", 1, "ki->ki", "FT"); es("Loo", 1, "lcki,lc->ki", "Lovoo", "t1")
perm("Lvv", 1, "ac->ac", "Fvv"); perm("Lvv", -1, "ac->ac", "FTv"); es("Lvv", 1, "kdac,kd->ac", "Lovvv", "t1")
# ---- singles
es("n1", -2, "ki,ka->ia", "FT", "t1")
es("n1", 1, "ac,ic->ia", "Fvv", "t1")
es("n1", -1, "ki,ka->ia", "Foo", "t1")
perm("Theta", 2, "kica->kica", "t2"); perm("Theta", -1, "ikca->kica", "t2")
es("n1", 1, "kc,kica->ia", "Fov", "Theta")
es("Q", 1, "kc,ic->ki", "Fov", "t1"); es("n1", 1, "ki,ka->ia", "Q", "t1")
perm("n1", 1, "ia->ia", "fov")
perm("Lph", 2, "kcai->kcai", "ovvo"); perm("Lph", -1, "kiac->kcai", "oovv")
es("n1", 1, "kcai,kc->ia", "Lph", "t1")
es("n1", 1, "kdac,ikcd->ia", "Lovvv", "tau")
es("n1", -1, "lcki,klac->ia", "Lovoo", "tau")
# ---- doubles: U collects everything that enters as X + X^T(ji,ba)
perm("G1", 1, "iacb->abic", "ovvv"); es("G1", -1, "kibc,ka->abic", "oovv", "t1")
es("U", 1, "abic,jc->ijab", "G1", "t1")
perm("G2", 1, "iajk->akij", "ovoo"); es("G2", 1, "kcai,jc->akij", "ovvo", "t1")
es("U", -1, "akij,kb->ijab", "G2", "t1")
perm("n2", 1, "iajb->ijab", "ovov")
perm("Woooo", 1, "kilj->klij", "oooo")
es("Woooo", 1, "lcki,jc->klij", "ovoo", "t1"); es("Woooo", 1, "kclj,ic->klij", "ovoo", "t1")
es("Woooo", 1, "kcld,ijcd->klij", "ovov", "tau")
es("n2", 1, "klij,klab->ijab", "Woooo", "tau")
ladder("n2", 1, "tau")
es("X", 1, "ijcd,kdac->ijka", "tau", "ovvv")              # t1 dressing of Wvvvv folded on the tau side
es("U", -1, "ijka,kb->ijab", "X", "t1")
es("U", 1, "ac,ijcb->ijab", "Lvv", "t2")
es("U", -1, "ki,kjab->ijab", "Loo", "t2")
# ring terms with FOUR (ov)^3 products (the factorisation of CcsdSolver::update_amps): in the ph layouts
#   Tph[i,a,l,d] = t2[i,l,a,d], Tpp[i,a,l,d] = t2[i,l,d,a], u = 2 Tph - Tpp, u~ = u - 2 t1(x)t1, Tp~ = Tpp + 2 t1(x)t1,
#   W1[i,a,k,c] = Wvoov[a,k,i,c] = base + 1/4 u~ L - 1/4 P,  W2[i,a,k,c] = Wvovo[a,k,c,i] = base - 1/2 P,  P = Tp~ ovov_t,
#   update = (W1 - W2/2) u - 1/2 A3 (as [i,a,j,b]) and -A3 (as [i,b,j,a]),  A3 = W2 Tpp.
perm("Tph", 1, "ilad->iald", "t2"); perm("Tpp", 1, "ilda->iald", "t2")
perm("uph", 2, "iald->iald", "Tph"); perm("uph", -1, "iald->iald", "Tpp")
perm("ut", 1, "iald->iald", "uph"); es("ut", -2, "id,la->iald", "t1", "t1")
perm("Tpt", 1, "iald->iald", "Tpp"); es("Tpt", 2, "id,la->iald", "t1", "t1")
perm("ovov_t", 1, "lckd->ldkc", "ovov")
es("P", 1, "iald,ldkc->iakc", "Tpt", "ovov_t")
perm("W1", 1, "kcai->iakc", "ovvo"); es("W1", 1, "kcad,id->iakc", "ovvv", "t1"); es("W1", -1, "kcli,la->iakc", "ovoo", "t1")
es("W1", 0.25, "iald,ldkc->iakc", "ut", "Lovov"); perm("W1", -0.25, "iakc->iakc", "P")
perm("W2", 1, "kiac->iakc", "oovv"); es("W2", 1, "kdac,id->iakc", "ovvv", "t1"); es("W2", -1, "lcki,la->iakc", "ovoo", "t1")
perm("W2", -0.5, "iakc->iakc", "P")
perm("Wc", 1, "iakc->iakc", "W1"); perm("Wc", -0.5, "iakc->iakc", "W2")
es("A3", 1, "ibkc,kcja->ibja", "W2", "Tpp")
perm("U", -1, "ibja->ijab", "A3")
es("Rf", 1, "iakc,kcjb->iajb", "Wc", "uph"); perm("Rf", -0.5, "iajb->iajb", "A3")
perm("U", 1, "iajb->ijab", "Rf")
perm("n2", 1, "ijab->ijab", "U"); perm("n2", 1, "jiba->ijab", "U")
# ---- energy (only its cotangents are used: E_bar = 1)
ENERGY = [("es", "E", 2.0, "ia,ia->", "fov", "t1"), ("es", "E", 1.0, "ijab,iajb->", "tau", "Lovov")]

OUTPUTS = ("n1", "n2")
T("vvvv_l", "vvvv", "virtual")        # cotangent of the ladder operand, vvvv_l_bar[a,b,c,d] = dL/d(ac|bd)


def bar(x):
    return x + "_bar"


def analyse():
    """zero-valued / t-dependent tensors, skip flags, backward program with its Lambda-iteration subset."""
    zero = {n for n, (_, k) in TENSORS.items() if k == "zero"}
    tdep = {"t1", "t2"}
    written = {}
    for st in FORWARD:
        ops = st[4:] if st[0] == "es" else (st[4],) if st[0] == "perm" else (st[3],)
        written.setdefault(st[1], []).append(ops)
    changed = True
    while changed:
        changed = False
        for dst, lst in written.items():
            if dst not in zero and all(any(o in zero for o in ops) for ops in lst):
                zero.add(dst); changed = True
            if dst not in tdep and any(o in tdep for ops in lst for o in ops):
                tdep.add(dst); changed = True
    fwd = []
    for st in FORWARD:
        ops = st[4:] if st[0] == "es" else (st[4],) if st[0] == "perm" else (st[3],)
        fwd.append((st, any(o in zero for o in ops)))          # (statement, skip_in_forward)
    bwd = []
    for st in reversed(FORWARD + ENERGY):
        kind, dst, coef = st[0], st[1], st[2]
        if kind == "es":
            subs, a, b = st[3], st[4], st[5]
            ins, out = subs.split("->"); ia, ib = ins.split(",")
            if dst == "E":      # E_bar = 1: the cotangent is a scaled copy of the other operand
                bwd.append(("perm", bar(a), coef, ib + "->" + ia, b, {"zero_src": b in zero}))
                bwd.append(("perm", bar(b), coef, ia + "->" + ib, a, {"zero_src": a in zero}))
                continue
            # d/da: a_bar[A] += coef * dst_bar[D] b[B];  vanishes identically when b is zero-valued
            bwd.append(("es", bar(a), coef, out + "," + ib + "->" + ia, bar(dst), b, {"zero_src": b in zero}))
            bwd.append(("es", bar(b), coef, ia + "," + out + "->" + ib, a, bar(dst), {"zero_src": a in zero}))
        elif kind == "perm":
            subs, a = st[3], st[4]
            ia, out = subs.split("->")
            bwd.append(("perm", bar(a), coef, out + "->" + ia, bar(dst), {"zero_src": False}))
        else:
            a = st[3]
            bwd.append(("ladder", bar(a), coef, bar(dst), {"zero_src": False}))
            bwd.append(("es", bar("vvvv_l"), coef, "ijab,ijcd->abcd", bar(dst), a, {"zero_src": False}))
    bwd = [s for s in bwd if not s[-1]["zero_src"]]
    # Lambda-iteration subset: statements whose destination feeds t1_bar / t2_bar
    needed = {bar("t1"), bar("t2")}
    changed = True
    while changed:
        changed = False
        for s in bwd:
            if s[1] in needed:
                srcs = [x for x in (s[4:-1] if s[0] != "ladder" else (s[3],)) if isinstance(x, str) and x.endswith("_bar")]
                for x in srcs:
                    if x not in needed:
                        needed.add(x); changed = True
    out = []
    for s in bwd:
        out.append(s[:-1] + ({"lam": s[1] in needed},))
    return fwd, out, zero, tdep


def all_tensors(bwd):
    names = dict(TENSORS)
    for s in bwd:
        base = s[1][:-4]
        if s[1] not in names:
            names[s[1]] = (TENSORS[base][0], "bar")
    for o in OUTPUTS:
        names.setdefault(bar(o), (TENSORS[o][0], "bar"))
    return names


def emit(path):
    fwd, bwd, zero, tdep = analyse()
    names = all_tensors(bwd)
    order = list(names)
    idx = {n: k for k, n in enumerate(order)}
    kind_code = {"t": 0, "int": 1, "zero": 2, "fwd": 3, "bar": 4, "virtual": 5}
    L = []
    L.append("// GENERATED by tools/gen_cc_lambda.py -- do not edit.  Tensor table, forward program (amplitude numerators) and its")
    L.append("// reverse sweep (Lambda equations + response densities).  See the generator for the equations and their source.")
    L.append("static const CcTensorDef kCcTensors[] = {")
    for n in order:
        sig, kind = names[n]
        L.append('  {"%s", "%s", %d, %d},' % (n, sig, kind_code[kind], 1 if n in zero else 0))
    L.append("};")

    def row(s, flag):
        if s[0] == "es":
            return '  {CC_ES, %d, %d, %d, %.17g, "%s", %d},' % (idx[s[1]], idx[s[4]], idx[s[5]], s[2], s[3], flag)
        if s[0] == "perm":
            return '  {CC_PERM, %d, %d, -1, %.17g, "%s", %d},' % (idx[s[1]], idx[s[4]], s[2], s[3], flag)
        return '  {CC_LADDER, %d, %d, -1, %.17g, "", %d},' % (idx[s[1]], idx[s[3]], s[2], flag)
    L.append("static const CcStmt kCcForward[] = {")
    for st, skip in fwd:
        L.append(row(st, 1 if skip else 0))          # flag 1: an operand is identically zero -> skipped
    L.append("};")
    L.append("static const CcStmt kCcBackward[] = {")
    for s in bwd:
        L.append(row(s[:-1], 1 if s[-1]["lam"] else 0))   # flag 1: needed in every Lambda iteration
    L.append("};")
    Path(path).write_text("\n".join(L) + "\n")
    return len(fwd), len(bwd), sum(1 for s in bwd if s[-1]["lam"])


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else str(Path(__file__).resolve().parent.parent / "quemb_amd" / "csrc" / "cc_lambda_program.inc")
    nf, nb, nl = emit(out)
    print("forward statements %d, backward %d (of which %d per Lambda iteration) -> %s" % (nf, nb, nl, out))
