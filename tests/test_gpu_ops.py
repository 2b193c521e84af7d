"""-m gpu: every device primitive of libqemb_hip (through the C ABI) against numpy on seeded inputs."""

import ctypes as C
import os

import numpy as np
import pytest

from quemb_amd import _lib
from quemb_amd._lib import DeviceBuffer, check, i64x4

pytestmark = pytest.mark.gpu


def _gemm(lib, A, B, Cm, alpha, beta, a_kc, b_kc, batch=1, cfg=-1):
    """A: (batch, M, K) logical; B: (batch, K, N) logical.  Storage chosen by a_kc / b_kc."""
    bsz, M, K = A.shape
    N = B.shape[2]
    As = A if a_kc else A.transpose(0, 2, 1)
    Bs = B.transpose(0, 2, 1) if b_kc else B
    As = np.ascontiguousarray(As)
    Bs = np.ascontiguousarray(Bs)
    dA, dB, dC = DeviceBuffer.from_numpy(As), DeviceBuffer.from_numpy(Bs), DeviceBuffer.from_numpy(Cm)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    lib.qemb_set_gemm_config(cfg)
    try:
        check(lib.qemb_op_gemm(M, N, K, alpha, dA.ptr, lda, int(a_kc), M * K, dB.ptr, ldb, int(b_kc), K * N, beta,
                               dC.ptr, N, M * N, bsz), "gemm")
    finally:
        lib.qemb_set_gemm_config(-1)
    return dC.numpy(Cm.shape)


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 4, 10, 11, 12, 33, 34, 35, 36, 37, 200, 204, 236, 237])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape", [(128, 128, 64), (400, 300, 200), (37, 53, 29), (441, 441, 441), (1, 220, 96), (130, 258, 18),
                                   (66, 130, 6), (64, 64, 16), (50, 70, 5)])      # the last three: a single k-tile (no second LDS buffer is ever filled)
def test_gemm_matches_numpy(qlib, cfg, a_kc, b_kc, shape):
    M, N, K = shape
    rng = np.random.default_rng(1234 + M + 7 * N + 13 * K)
    A = rng.standard_normal((2, M, K))
    B = rng.standard_normal((2, K, N))
    C0 = rng.standard_normal((2, M, N))
    alpha, beta = 0.75, -0.5
    got = _gemm(qlib, A, B, C0, alpha, beta, a_kc, b_kc, cfg=cfg)
    ref = alpha * np.einsum("bmk,bkn->bmn", A, B) + beta * C0
    err = np.abs(got - ref).max()
    assert err < 1e-11 * max(1.0, K), (shape, cfg, a_kc, b_kc, err)


def test_gemm_identity_asymmetric(qlib):
    """A = I with an asymmetric B catches swapped C-layout maps (guide section 3)."""
    n = 64
    A = np.eye(n)[None]
    B = (np.arange(n * n, dtype=np.float64).reshape(n, n) * 1.0)[None]
    got = _gemm(qlib, A, B, np.zeros((1, n, n)), 1.0, 0.0, 1, 0)
    assert np.array_equal(got[0], B[0])


def test_gemm_beta_zero_ignores_nan(qlib):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((1, 70, 40)); B = rng.standard_normal((1, 40, 90))
    C0 = np.full((1, 70, 90), np.nan)
    got = _gemm(qlib, A, B, C0, 1.0, 0.0, 1, 1)
    assert np.allclose(got, A @ B, atol=1e-12)


@pytest.mark.parametrize("perm", [(0, 1, 2, 3), (0, 2, 1, 3), (1, 0, 3, 2), (3, 2, 1, 0), (2, 3, 0, 1), (0, 1, 3, 2), (1, 3, 0, 2)])
def test_copy4_permutations(qlib, perm):
    rng = np.random.default_rng(7)
    dims = (6, 9, 33, 41)
    x = rng.standard_normal(dims)
    out0 = rng.standard_normal(tuple(dims[p] for p in perm))
    alpha, beta = 1.5, 0.25
    ref = alpha * x.transpose(perm) + beta * out0
    # loop over the INPUT dims; output strides follow the permuted layout
    out_shape = ref.shape
    ostr = np.zeros(4, dtype=np.int64)
    st = np.array([int(np.prod(out_shape[k + 1:])) for k in range(4)])
    for k, p in enumerate(perm):
        ostr[p] = st[k]
    istr = [int(np.prod(dims[k + 1:])) for k in range(4)]
    dx, do = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(out0)
    check(qlib.qemb_op_copy4(i64x4(dims), dx.ptr, i64x4(istr), do.ptr, i64x4(ostr), alpha, beta))
    got = do.numpy(out_shape)
    assert np.allclose(got, ref, atol=1e-14)


def test_copy4_block_extract(qlib):
    rng = np.random.default_rng(8)
    n, o = 11, 4
    v = n - o
    M = rng.standard_normal((n, n, n, n))
    dM = DeviceBuffer.from_numpy(M)
    dO = DeviceBuffer(o * v * o * v)
    s = [n ** 3, n ** 2, n, 1]
    off = o * n * n + o  # [i, o+a, j, o+b]
    check(qlib.qemb_op_copy4(i64x4((o, v, o, v)), dM.at(off), i64x4(s), dO.ptr, i64x4((v * o * v, o * v, v, 1)), 1.0, 0.0))
    assert np.array_equal(dO.numpy((o, v, o, v)), M[:o, o:, :o, o:])


def test_outer4_and_denominators(qlib):
    rng = np.random.default_rng(9)
    o, v = 5, 7
    t1 = rng.standard_normal((o, v)); t2 = rng.standard_normal((o, o, v, v))
    d1, d2 = DeviceBuffer.from_numpy(t1), DeviceBuffer.from_numpy(t2)
    check(qlib.qemb_op_outer4(i64x4((o, o, v, v)), d1.ptr, v, 1, d1.ptr, v, 1, d2.ptr, i64x4((o * v * v, v * v, v, 1)), 1.0, 1.0))
    tau = t2 + np.einsum("ia,jb->ijab", t1, t1)
    assert np.allclose(d2.numpy(t2.shape), tau, atol=1e-14)
    eo = np.sort(rng.standard_normal(o)) - 3; ev = np.sort(rng.standard_normal(v)) + 3
    deo, dev = DeviceBuffer.from_numpy(eo), DeviceBuffer.from_numpy(ev)
    check(qlib.qemb_op_div_denom(d2.ptr, o, o, v, v, deo.ptr, deo.ptr, dev.ptr, dev.ptr))
    den = eo[:, None, None, None] + eo[None, :, None, None] - ev[None, None, :, None] - ev[None, None, None, :]
    assert np.allclose(d2.numpy(t2.shape), tau / den, rtol=1e-14)
    check(qlib.qemb_op_div_denom(d1.ptr, o, 1, v, 1, deo.ptr, None, dev.ptr, None))
    assert np.allclose(d1.numpy(t1.shape), t1 / (eo[:, None] - ev[None, :]), rtol=1e-14)


def test_reductions(qlib):
    rng = np.random.default_rng(10)
    for n in (1, 1000, 1 << 20, 3_000_001):
        x = rng.standard_normal(n); y = rng.standard_normal(n)
        dx, dy, do = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(y), DeviceBuffer(2)
        check(qlib.qemb_op_dot(n, dx.ptr, dy.ptr, do.ptr))
        check(qlib.qemb_op_absmax(n, dx.ptr, do.at(1)))
        r = do.numpy()
        assert abs(r[0] - x @ y) < 1e-9 * max(1.0, np.sqrt(n))
        assert r[1] == np.abs(x).max()
        check(qlib.qemb_op_dot(n, dx.ptr, dy.ptr, do.ptr))
        assert do.numpy()[0] == r[0], "reduction must be run-to-run deterministic"


def test_gemv_and_contract_mid(qlib):
    rng = np.random.default_rng(11)
    n = 23
    E = rng.standard_normal((n, n, n, n)); D = rng.standard_normal((n, n))
    dE, dD = DeviceBuffer.from_numpy(E), DeviceBuffer.from_numpy(D)
    dJ, dK = DeviceBuffer(n * n), DeviceBuffer(n * n)
    check(qlib.qemb_op_gemv_rows(n * n, n * n, dE.ptr, n * n, dD.ptr, dJ.ptr, 1.0, 0.0))
    assert np.allclose(dJ.numpy((n, n)), np.einsum("pqrs,rs->pq", E, D), atol=1e-11)
    check(qlib.qemb_op_contract_mid(n, n * n, n, dE.ptr, dD.ptr, dK.ptr, n, 1.0, 0.0))
    assert np.allclose(dK.numpy((n, n)), np.einsum("pqsr,qs->pr", E, D), atol=1e-11)


def test_gemv_rows_batched(qlib):
    rng = np.random.default_rng(12)
    o, v = 5, 13
    T = rng.standard_normal((o, v * v, v)); x = rng.standard_normal((o, v)); y0 = rng.standard_normal(v * v)
    dT, dx, dy = DeviceBuffer.from_numpy(T), DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(y0)
    check(qlib.qemb_op_gemv_rows_batched(v * v, v, o, dT.ptr, v, v * v * v, dx.ptr, v, dy.ptr, 0.5, 2.0))
    assert np.allclose(dy.numpy(), 0.5 * np.einsum("brc,bc->r", T, x) + 2.0 * y0, atol=1e-12)


def _sym_eri(n, rng):
    B = rng.standard_normal((2 * n, n, n)); B = B + B.transpose(0, 2, 1)
    return np.einsum("Ppq,Prs->pqrs", B, B)


@pytest.mark.parametrize("n", [9, 32, 45])     # n >= 32 takes the LDS-tiled unpack kernels
def test_pack_unpack(qlib, n):
    rng = np.random.default_rng(12)
    npair = n * (n + 1) // 2
    eri = _sym_eri(n, rng)
    il = np.tril_indices(n)
    s4 = eri[il][:, il[0], il[1]]
    d4, d1 = DeviceBuffer.from_numpy(s4), DeviceBuffer(n ** 4)
    check(qlib.qemb_op_unpack_s4(n, d4.ptr, d1.ptr))
    assert np.array_equal(d1.numpy((n,) * 4), eri)
    d4b = DeviceBuffer(npair * npair)
    check(qlib.qemb_op_pack_s4(n, d1.ptr, d4b.ptr))
    assert np.array_equal(d4b.numpy((npair, npair)), s4)
    s8 = s4[np.tril_indices(npair)]
    d8 = DeviceBuffer.from_numpy(s8)
    check(qlib.qemb_op_unpack_s8_to_s4(n, d8.ptr, d4b.ptr))
    assert np.array_equal(d4b.numpy((npair, npair)), s4)
    rows = 5
    P = rng.standard_normal((rows, npair))
    dP, dF = DeviceBuffer.from_numpy(P), DeviceBuffer(rows * n * n)
    check(qlib.qemb_op_unpack_tril_rows(rows, n, dP.ptr, dF.ptr))
    F = dF.numpy((rows, n, n))
    assert np.array_equal(F[:, il[0], il[1]], P) and np.array_equal(F, F.transpose(0, 2, 1))
    dP2 = DeviceBuffer(rows * npair)
    check(qlib.qemb_op_pack_tril_rows(rows, n, dF.ptr, dP2.ptr))
    assert np.array_equal(dP2.numpy((rows, npair)), P)


@pytest.mark.parametrize("n", [1, 31, 100])
def test_mirror_lower(qlib, n):
    rng = np.random.default_rng(5 + n)
    A = rng.standard_normal((n, n + 3))
    dA = DeviceBuffer.from_numpy(A)
    check(qlib.qemb_op_mirror_lower(n, dA.ptr, n + 3))
    B = dA.numpy((n, n + 3))
    ref = A.copy(); sq = np.tril(A[:, :n]); ref[:, :n] = sq + np.tril(sq, -1).T
    assert np.array_equal(B, ref)


def test_pair_packed_transform_helpers(qlib):
    """dev_pack_pair_rows / dev_unpack_tril_pair_rows / dev_extract_pf / dev_extract_pf_t / dev_ladder_pack_vvvv_pf against numpy indexing."""
    rng = np.random.default_rng(77)
    for n, o in ((11, 3), (37, 5)):                   # n >= 32 takes the LDS-tiled fused unpack
        v = n - o
        npair = n * (n + 1) // 2
        il = np.tril_indices(n)
        ncols = 13
        A = rng.standard_normal((n * n, ncols))
        dA, dO = DeviceBuffer.from_numpy(A), DeviceBuffer(npair * ncols)
        check(qlib.qemb_op_pack_pair_rows(n, ncols, dA.ptr, dO.ptr))
        assert np.array_equal(dO.numpy((npair, ncols)), A.reshape(n, n, ncols)[il])
        eri = _sym_eri(n, rng)
        # rows (x,y) all n*n, columns packed pairs -> keep x >= y rows and unpack the columns in one pass
        Xin = np.ascontiguousarray(eri[:, :, il[0], il[1]]).reshape(n * n, npair)
        dX, dF = DeviceBuffer.from_numpy(Xin), DeviceBuffer(npair * n * n)
        check(qlib.qemb_op_unpack_tril_pair_rows(n, n, dX.ptr, dF.ptr))
        Mp = eri[il]                                                # Mp[P(p,q)][r][s]
        assert np.array_equal(dF.numpy((npair, n, n)), Mp)
        dM = dF
        for (p0, q0, r0, s0, sp, sq, sr, ss) in [(0, 0, 0, 0, o, o, o, o), (0, o, o, 0, o, v, v, o), (0, o, o, o, o, v, v, v), (o, o, o, o, v, v, v, v)]:
            dB = DeviceBuffer(sp * sq * sr * ss)
            check(qlib.qemb_op_extract_pf(n, dM.ptr, p0, q0, r0, s0, sp, sq, sr, ss, dB.ptr))
            assert np.array_equal(dB.numpy((sp, sq, sr, ss)), eri[p0:p0 + sp, q0:q0 + sq, r0:r0 + sr, s0:s0 + ss])
        # T[P(r,s)][c][x] = (c x|r s) -> out[x][r][s][c]
        dB = DeviceBuffer(v * o * v * 4)
        check(qlib.qemb_op_extract_pf_t(n, dM.ptr, o, 0, o, 0, v, o, v, 4, dB.ptr))
        assert np.array_equal(dB.numpy((v, o, v, 4)), eri[:4, o:, :o, o:].transpose(1, 2, 3, 0))
        npv, nmv = v * (v + 1) // 2, v * (v - 1) // 2
        ldp, ldm = npv + (npv & 1), nmv + (nmv & 1)
        dVp, dVm = DeviceBuffer(npv * ldp), DeviceBuffer(max(nmv, 1) * ldm)
        dVp2, dVm2 = DeviceBuffer(npv * ldp), DeviceBuffer(max(nmv, 1) * ldm)
        check(qlib.qemb_op_ladder_pack_vvvv_pf(n, o, dM.ptr, dVp.ptr, ldp, dVm.ptr, ldm))
        dE = DeviceBuffer.from_numpy(eri)
        check(qlib.qemb_op_ladder_pack_vvvv(n, o, dE.ptr, dVp2.ptr, ldp, dVm2.ptr, ldm))
        assert np.array_equal(dVp.numpy((npv, ldp)), dVp2.numpy((npv, ldp)))
        assert np.array_equal(dVm.numpy((nmv, ldm)), dVm2.numpy((nmv, ldm)))


@pytest.mark.parametrize("n", [2, 7, 42, 57, 80, 81, 90, 96, 97, 131, 220])
def test_jacobi_eigh(qlib, n):
    rng = np.random.default_rng(13 + n)
    A = rng.standard_normal((n, n)); A = A + A.T
    if n >= 8:  # plant a +/- pair and a degenerate pair
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        w = rng.standard_normal(n); w[0], w[1] = 1.5, -1.5; w[2] = w[3] = 0.25
        A = (Q * w) @ Q.T
    dA, dw, dV = DeviceBuffer.from_numpy(A), DeviceBuffer(n), DeviceBuffer(n * n)
    sw = C.c_int(0)
    check(qlib.qemb_op_jacobi_eigh(n, dA.ptr, dw.ptr, dV.ptr, C.byref(sw)))
    w, V = dw.numpy(), dV.numpy((n, n))
    wr = np.linalg.eigvalsh(A)
    scale = np.abs(wr).max()
    assert np.abs(w - wr).max() < 1e-12 * scale
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
    assert np.abs(A @ V - V * w).max() < 1e-11 * scale


def check_small_copies(lib):
    """host <-> device copies around the size of the pinned slots (256 KB) and more asynchronous uploads in a row than there are slots: every byte arrives, a host
    buffer may be overwritten as soon as the upload call has returned"""
    rng = np.random.default_rng(5)
    slot = 256 * 1024 // 8
    sizes = [1, 7, 1000, slot - 1, slot, slot + 1, 3 * slot]
    for n in sizes:                                   # the waiting pair
        a = rng.standard_normal(n)
        d = DeviceBuffer.from_numpy(a, lib=lib)
        assert np.array_equal(d.numpy(), a)
        d.free()
    bufs, want = [], []
    scratch = np.empty(slot)
    for k in range(28):                               # > 3 x the ring of eight slots, mixed sizes, ONE host buffer reused for all of them
        n = [5, 4096, slot, slot - 3][k % 4]
        scratch[:n] = rng.standard_normal(n)
        want.append(scratch[:n].copy())
        d = DeviceBuffer(n, lib=lib)
        check(lib.qemb_h2d_async(d.ptr, scratch.ctypes.data, n * 8), "qemb_h2d_async", lib)
        scratch[:n] = -1.0                            # the source is free on return
        bufs.append(d)
    big = rng.standard_normal(2 * slot + 5)           # beyond a slot: the waiting path behind the same call
    dbig = DeviceBuffer(big.size, lib=lib)
    check(lib.qemb_h2d_async(dbig.ptr, big.ctypes.data, big.size * 8), "qemb_h2d_async", lib)
    for d, w in zip(bufs, want):
        assert np.array_equal(d.numpy(), w)
        d.free()
    assert np.array_equal(dbig.numpy(), big)
    dbig.free()


def test_small_copies_through_pinned_slots(qlib):
    check_small_copies(qlib)


def check_fused_scf_ops(lib, sizes):
    """The fused steps of the SCF cycle of small fragments (linalg_f64.hip; reference molbe/helper.py:73-151) against NumPy: packed density, Fock + energy + commutator,
    eigenproblem in a rotated basis with back-rotation, copy and density."""
    nmax = lib.qemb_op_scf_fused_max()
    for n in sizes:
        if n > nmax:
            continue
        rng = np.random.default_rng(100 + n)
        sym = lambda a: 0.5 * (a + a.T)
        h, J, K = (sym(rng.standard_normal((n, n))) for _ in range(3))
        nocc = max(1, n // 3)
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        D = 2.0 * Q[:, :nocc] @ Q[:, :nocc].T + 1e-3 * rng.standard_normal((n, n))      # (not exactly symmetric: the packed density must add both halves)
        dh, dJ, dK, dD = (DeviceBuffer.from_numpy(x, lib=lib) for x in (h, J, K, D))
        dF, dE, dS = DeviceBuffer(n * n, lib=lib), DeviceBuffer(n * n, lib=lib), DeviceBuffer(2, lib=lib)
        check(lib.qemb_op_scf_fock_small(n, dh.ptr, dJ.ptr, dK.ptr, dD.ptr, dF.ptr, dE.ptr, dS.ptr), lib=lib)
        F = h + J - 0.5 * K
        err = F @ D - D @ F
        assert np.abs(dF.numpy((n, n)) - F).max() < 1e-14 * max(1.0, np.abs(F).max())
        assert np.abs(dE.numpy((n, n)) - err).max() < 1e-12 * max(1.0, np.abs(err).max())
        sc = dS.numpy()
        assert abs(sc[0] - ((h + F) * D).sum()) < 1e-11 * max(1.0, abs(((h + F) * D).sum())) and abs(sc[1] - (err * err).sum()) < 1e-11 * (err * err).sum()
        # packed density
        dP = DeviceBuffer(n * (n + 1) // 2, lib=lib)
        check(lib.qemb_op_pack_density_sym(n, dD.ptr, dP.ptr), lib=lib)
        il = np.tril_indices(n)
        want = np.where(il[0] == il[1], D[il], D[il] + D.T[il])
        assert np.abs(dP.numpy() - want).max() == 0.0
        # eigenproblem in the basis Cp, with and without a basis, C2 aliasing Cp, density of the lowest nocc
        for with_basis in (True, False):
            Cp = np.linalg.qr(rng.standard_normal((n, n)))[0]
            Fd = sym(rng.standard_normal((n, n))) + np.diag(3.0 * np.arange(n))
            dFd, dCp = DeviceBuffer.from_numpy(Fd, lib=lib), DeviceBuffer.from_numpy(Cp, lib=lib)
            dw, dC, dDm = DeviceBuffer(n, lib=lib), DeviceBuffer(n * n, lib=lib), DeviceBuffer(n * n, lib=lib)
            sw = C.c_int(0)
            check(lib.qemb_op_jacobi_eigh_in_basis(n, dFd.ptr, dCp.ptr if with_basis else None, dw.ptr, dC.ptr, dCp.ptr, nocc, dDm.ptr, 1e-10, C.byref(sw)), lib=lib)
            w, Cm, C2, Dm = dw.numpy(), dC.numpy((n, n)), dCp.numpy((n, n)), dDm.numpy((n, n))
            wr = np.linalg.eigvalsh(Fd)
            scale = np.abs(wr).max()
            assert np.abs(w - wr).max() < 1e-12 * scale and sw.value >= 1
            assert np.abs(Cm.T @ Cm - np.eye(n)).max() < 1e-12
            assert np.abs(Fd @ Cm - Cm * w).max() < 1e-11 * scale
            assert np.array_equal(C2, Cm)                                                   # the copy (written over Cp)
            assert np.abs(Dm - 2.0 * Cm[:, :nocc] @ Cm[:, :nocc].T).max() < 1e-13
            for b in (dFd, dCp, dw, dC, dDm):
                b.free()
        for b in (dh, dJ, dK, dD, dF, dE, dS, dP):
            b.free()


def test_fused_scf_ops(qlib):
    check_fused_scf_ops(qlib, (2, 7, 24, 41, 42, 57, 64, 79, 80))


def test_jacobi_eigh_projector_spectrum(qlib):
    """The Schmidt case: environment block of an idempotent 1-RDM (eigenvalues 0, 1 and a few in between)."""
    rng = np.random.default_rng(14)
    N, nocc, nf = 60, 20, 6
    Cm = np.linalg.qr(rng.standard_normal((N, N)))[0][:, :nocc]
    D = Cm @ Cm.T
    Denv = D[nf:, nf:]
    n = N - nf
    dA, dw, dV = DeviceBuffer.from_numpy(Denv), DeviceBuffer(n), DeviceBuffer(n * n)
    check(qlib.qemb_op_jacobi_eigh(n, dA.ptr, dw.ptr, dV.ptr, None))
    w, V = dw.numpy(), dV.numpy((n, n))
    wr, Vr = np.linalg.eigh(Denv)
    assert np.abs(w - wr).max() < 1e-13
    sel = (np.abs(w) > 1e-10) & (np.abs(w) < 1 - 1e-10)
    selr = (np.abs(wr) > 1e-10) & (np.abs(wr) < 1 - 1e-10)
    assert sel.sum() == selr.sum() == nf
    Pg, Pr = V[:, sel] @ V[:, sel].T, Vr[:, selr] @ Vr[:, selr].T
    assert np.abs(Pg - Pr).max() < 1e-11


@pytest.mark.parametrize("m,n", [(50, 6), (300, 22), (40, 40)])
def test_jacobi_svd(qlib, m, n):
    rng = np.random.default_rng(15 + m)
    G = rng.standard_normal((m, n))
    if n >= 6:
        G[:, 1] = G[:, 0]  # rank deficient: one exact zero singular value
    dG, ds, dU, dV = DeviceBuffer.from_numpy(G), DeviceBuffer(n), DeviceBuffer(m * n), DeviceBuffer(n * n)
    check(qlib.qemb_op_jacobi_svd(m, n, dG.ptr, ds.ptr, dU.ptr, dV.ptr, None))
    s, U, V = ds.numpy(), dU.numpy((m, n)), dV.numpy((n, n))
    sr = np.linalg.svd(G, compute_uv=False)
    assert np.abs(s - sr).max() < 1e-12 * sr.max()
    assert np.abs((U * s) @ V.T - G).max() < 1e-12 * sr.max()
    r = int((sr > 1e-10 * sr.max()).sum())
    assert np.abs(U[:, :r].T @ U[:, :r] - np.eye(r)).max() < 1e-12


@pytest.mark.parametrize("n", [5, 32, 77, 300])
def test_cholesky_and_tri_inverse(qlib, n):
    rng = np.random.default_rng(16 + n)
    X = rng.standard_normal((n, n)); A = X @ X.T + n * np.eye(n)
    dA, dI = DeviceBuffer.from_numpy(A), DeviceBuffer(n * n)
    check(qlib.qemb_op_cholesky_lower(n, dA.ptr))
    L = dA.numpy((n, n))
    Lr = np.linalg.cholesky(A)
    assert np.abs(L - Lr).max() < 1e-11 * np.abs(Lr).max()
    check(qlib.qemb_op_tri_inverse_lower(n, dA.ptr, dI.ptr))
    Li = dI.numpy((n, n))
    assert np.abs(Li @ L - np.eye(n)).max() < 1e-11
    assert np.abs(np.triu(Li, 1)).max() == 0.0


def test_cholesky_rejects_indefinite(qlib):
    A = np.eye(8); A[3, 3] = -1.0
    dA = DeviceBuffer.from_numpy(A)
    assert qlib.qemb_op_cholesky_lower(8, dA.ptr) == -5


@pytest.mark.parametrize("shape", [(20, 200, 80000), (20, 20, 50000), (200, 200, 8000), (400, 400, 40000), (20, 22, 80001)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0)])
def test_gemm_split_k(qlib, shape, a_kc, b_kc):
    """few output tiles + long K: the dispatcher splits K over workgroups and reduces deterministically."""
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    got = _gemm(qlib, A, B, C0, 0.5, 2.0, a_kc, b_kc)
    ref = 0.5 * (A @ B) + 2.0 * C0
    assert np.abs(got - ref).max() < 1e-12 * K
    again = _gemm(qlib, A, B, C0, 0.5, 2.0, a_kc, b_kc)
    assert np.array_equal(got, again)
    qlib.qemb_set_gemm_splitk(0)
    try:
        plain = _gemm(qlib, A, B, C0, 0.5, 2.0, a_kc, b_kc)
    finally:
        qlib.qemb_set_gemm_splitk(1)
    assert np.abs(got - plain).max() < 1e-12 * K


def test_mfma_f64_peak_calibration(qlib):
    t = C.c_double()
    check(qlib.qemb_mfma_f64_peak(20000, 2, C.byref(t)))
    assert 20.0 < t.value < 200.0


@pytest.mark.parametrize("cfg,M", [(10, 210), (11, 100), (12, 64), (10, 224)])
@pytest.mark.parametrize("ks", [0, 3])
def test_gemm_single_m_tile_configs(qlib, cfg, M, ks):
    rng = np.random.default_rng(cfg + M)
    N, K = 1000, 1536
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    qlib.qemb_set_gemm_ksplit(ks)
    try:
        got = _gemm(qlib, A, B, C0, 1.0, 0.0, 1, 1, cfg=cfg)
    finally:
        qlib.qemb_set_gemm_ksplit(0)
    assert np.abs(got - A @ B).max() < 1e-10


@pytest.mark.parametrize("o,v", [(3, 5), (6, 17), (4, 40)])
def test_pm_packed_ladder_equals_dense(qlib, o, v):
    """pack (+/-) operands, two GEMMs over pairs, scatter == dense sum_cd (ac|bd) tau_ijcd for every (i,j,a,b)."""
    rng = np.random.default_rng(o * 100 + v)
    n = o + v
    Bm = rng.standard_normal((2 * n, n, n)); Bm = Bm + Bm.transpose(0, 2, 1)
    M = np.einsum("Ppq,Prs->pqrs", Bm, Bm)
    tau = rng.standard_normal((o, o, v, v)); tau = tau + tau.transpose(1, 0, 3, 2)
    npv, nmv, npo, nmo = v * (v + 1) // 2, v * (v - 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2
    ldp, ldm = npv + (npv & 1), nmv + (nmv & 1)
    dM, dT = DeviceBuffer.from_numpy(M), DeviceBuffer.from_numpy(tau)
    dVp, dVm = DeviceBuffer(npv * ldp), DeviceBuffer(max(nmv, 1) * ldm)
    dTp, dTm = DeviceBuffer(npo * ldp), DeviceBuffer(max(nmo, 1) * ldm)
    dRp, dRm = DeviceBuffer(npo * ldp), DeviceBuffer(max(nmo, 1) * ldm)
    check(qlib.qemb_op_ladder_pack_vvvv(n, o, dM.ptr, dVp.ptr, ldp, dVm.ptr, ldm))
    check(qlib.qemb_op_ladder_pack_tau(o, v, dT.ptr, dTp.ptr, ldp, dTm.ptr, ldm))
    vv = M[o:, o:, o:, o:]
    il, sl = np.tril_indices(v), np.tril_indices(v, -1)
    Vp = dVp.numpy((npv, ldp))[:, :npv]
    ref_p = (vv.transpose(0, 2, 1, 3) + vv.transpose(0, 2, 3, 1))[il[0], il[1]][:, il[0], il[1]]     # (ac|bd)+(ad|bc) at [ab,cd]
    assert np.allclose(Vp, ref_p, atol=1e-12)
    check(qlib.qemb_op_gemm(npo, npv, ldp, 1.0, dTp.ptr, ldp, 1, 0, dVp.ptr, ldp, 1, 0, 0.0, dRp.ptr, ldp, 0, 1))
    check(qlib.qemb_op_gemm(nmo, nmv, ldm, 1.0, dTm.ptr, ldm, 1, 0, dVm.ptr, ldm, 1, 0, 0.0, dRm.ptr, ldm, 0, 1))
    t2 = rng.standard_normal((o, o, v, v))
    d2 = DeviceBuffer.from_numpy(t2)
    check(qlib.qemb_op_ladder_scatter_pm(o, v, dRp.ptr, ldp, dRm.ptr, ldm, d2.ptr))
    ref = t2 + np.einsum("acbd,ijcd->ijab", vv, tau)
    assert np.abs(d2.numpy(t2.shape) - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("n", [3, 17, 70])
def test_k_from_pairs_and_unpack(qlib, n):
    """K[p,r] = sum_qs (pq|rs) D[q,s] from the half-unpacked pair-row tensor == the dense contraction."""
    rng = np.random.default_rng(n)
    eri = _sym_eri(n, rng)
    il = np.tril_indices(n)
    npair = len(il[0])
    s4 = eri[il][:, il[0], il[1]]
    D = rng.standard_normal((n, n))                     # not symmetric on purpose
    d4, dH = DeviceBuffer.from_numpy(s4), DeviceBuffer(npair * n * n)
    check(qlib.qemb_op_unpack_tril_rows(npair, n, d4.ptr, dH.ptr))
    assert np.array_equal(dH.numpy((npair, n, n)), eri[il])
    dD, dK = DeviceBuffer.from_numpy(D), DeviceBuffer(n * n)
    check(qlib.qemb_op_k_from_pairs(n, dH.ptr, dD.ptr, dK.ptr))
    ref = np.einsum("pqrs,qs->pr", eri, D)
    assert np.abs(dK.numpy((n, n)) - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n", [1, 3, 17, 70, 131, 220, 300])
def test_jk_from_packed_block(qlib, n):
    """J and K in one pass over the 4-fold packed block (scf.cpp build_jk) == the dense contractions; n = 131 / 220 / 300 span several
    LDS chunks and (300) two column owners per thread; a non-symmetric D on purpose; K only; run-to-run bitwise identical."""
    rng = np.random.default_rng(1000 + n)
    il = np.tril_indices(n)
    npair = len(il[0])
    if n <= 70:
        eri = _sym_eri(n, rng)
        s4 = np.ascontiguousarray(eri[il][:, il[0], il[1]])
    else:                       # (pq|rs) = sum_P B B at sizes where the dense tensor is too big for einsum on the host
        Bp = rng.standard_normal((5, npair))
        s4 = Bp.T @ Bp
    D = rng.standard_normal((n, n))
    Ds = D + D.T - np.diag(np.diag(D))
    Dp = np.ascontiguousarray(Ds[il])
    d4, dD, dDp = DeviceBuffer.from_numpy(s4), DeviceBuffer.from_numpy(D), DeviceBuffer.from_numpy(Dp)
    dJ, dK, dK2 = DeviceBuffer(npair), DeviceBuffer(n * n), DeviceBuffer(n * n)
    check(qlib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, dDp.ptr, dJ.ptr, dK.ptr))
    check(qlib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, None, None, dK2.ptr))
    K, K2, Jp = dK.numpy((n, n)), dK2.numpy((n, n)), dJ.numpy((npair,))
    assert np.array_equal(K, K2)
    Jref = s4 @ Dp
    assert np.abs(Jp - Jref).max() < 1e-11 * max(1.0, np.abs(Jref).max())
    if n <= 70:
        Kref = np.einsum("pqrs,qs->pr", eri, D)
    else:                       # K[p,r] = sum_P (B_P D B_P^T)[p,r] with the symmetric matrices B_P
        Bf = np.zeros((5, n, n)); Bf[:, il[0], il[1]] = Bp; Bf = Bf + Bf.transpose(0, 2, 1) - np.einsum("pii->pi", Bf)[:, :, None] * np.eye(n)
        Kref = np.einsum("Ppq,qs,Prs->pr", Bf, D, Bf, optimize=True)
    assert np.abs(K - Kref).max() < 1e-11 * max(1.0, np.abs(Kref).max())
    check(qlib.qemb_op_jk_from_packed(n, d4.ptr, dD.ptr, dDp.ptr, dJ.ptr, dK2.ptr))
    assert np.array_equal(dK2.numpy((n, n)), K) and np.array_equal(dJ.numpy((npair,)), Jp)


def test_jk_from_packed_block_argument_errors(qlib):
    from quemb_amd._lib import QembError
    d = DeviceBuffer(16)
    with pytest.raises(QembError):
        check(qlib.qemb_op_jk_from_packed(1025, d.ptr, d.ptr, None, None, d.ptr))       # n > 1024: the pair-row path is the one to use
    with pytest.raises(QembError):
        check(qlib.qemb_op_jk_from_packed(2, d.ptr, d.ptr, None, d.ptr, d.ptr))          # Jp without Dp
    with pytest.raises(QembError):
        check(qlib.qemb_op_jk_from_packed(2, d.ptr, d.ptr, None, None, None))            # nothing to compute


def test_pm_pair_packing_roundtrip_and_lincomb(qlib):
    rng = np.random.default_rng(9)
    rows, v, o, ncols = 5, 7, 4, 6
    A = rng.standard_normal((rows, v, v))
    npv, nmv = v * (v + 1) // 2, v * (v - 1) // 2
    ldp, ldm = npv + (npv & 1), nmv + (nmv & 1)
    dA, dP, dM = DeviceBuffer.from_numpy(A), DeviceBuffer(rows * ldp), DeviceBuffer(rows * ldm)
    check(qlib.qemb_op_pack_pm_cols(rows, v, dA.ptr, dP.ptr, ldp, dM.ptr, ldm))
    P, M = dP.numpy((rows, ldp)), dM.numpy((rows, ldm))
    il = np.tril_indices(v); ils = np.tril_indices(v, -1)
    assert np.array_equal(P[:, :npv], (A + A.transpose(0, 2, 1))[:, il[0], il[1]]) and not P[:, npv:].any()
    assert np.array_equal(M[:, :nmv], (A - A.transpose(0, 2, 1))[:, ils[0], ils[1]]) and not M[:, nmv:].any()
    npo, nmo = o * (o + 1) // 2, o * (o - 1) // 2
    Xp, Xm = rng.standard_normal((npo, ncols)), rng.standard_normal((nmo, ncols))
    dXp, dXm, dO = DeviceBuffer.from_numpy(Xp), DeviceBuffer.from_numpy(Xm), DeviceBuffer(o * o * ncols)
    check(qlib.qemb_op_scatter_pm_rows(o, ncols, dXp.ptr, dXm.ptr, dO.ptr))
    out = dO.numpy((o, o, ncols))
    for i in range(o):
        for j in range(i + 1):
            p = Xp[i * (i + 1) // 2 + j]
            if i == j:
                assert np.array_equal(out[i, i], p)
            else:
                m = Xm[i * (i - 1) // 2 + j]
                assert np.array_equal(out[i, j], p + m) and np.array_equal(out[j, i], p - m)
    x, y, z = rng.standard_normal(1000), rng.standard_normal(1000), rng.standard_normal(1000)
    dx, dy, dz = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(y), DeviceBuffer.from_numpy(z)
    check(qlib.qemb_op_lincomb2(1000, 2.0, dx.ptr, -0.5, dy.ptr, 3.0, dz.ptr))
    assert np.abs(dz.numpy() - (2.0 * x - 0.5 * y + 3.0 * z)).max() < 1e-14


@pytest.mark.parametrize("o,v", [(1, 1), (3, 5), (4, 33), (7, 70), (4, 36), (8, 50), (2, 200), (6, 40), (3, 34), (5, 96)])
def test_ccsd_single_pass_kernels(qlib, o, v):
    """The fused element-wise kernels of the amplitude update against NumPy, at sizes that are not multiples of the 32 x 32 tiles; the last
    three of the first seven have o v a multiple of 16, where ph_layouts shifts its tile columns onto 128-byte lines (a different shift per (k, j)); even
    n_virt >= 32 takes the 16-byte-access kernel (32 x 64 tiles)."""
    rng = np.random.default_rng(100 * o + v)
    t2 = rng.standard_normal((o, o, v, v)); t1 = rng.standard_normal((o, v))
    d2, d1 = DeviceBuffer.from_numpy(t2), DeviceBuffer.from_numpy(t1)
    outs = [DeviceBuffer(o * o * v * v) for _ in range(6)]
    check(qlib.qemb_op_ccsd_ph_layouts(o, v, d2.ptr, d1.ptr, *[b.ptr for b in outs]))
    T, Tp, S, Ut, Tpt = [b.numpy((o, v, o, v)) for b in outs[:5]]
    Th = outs[5].numpy((o, o, v, v))
    T_ref = t2.transpose(0, 2, 1, 3); Tp_ref = t2.transpose(0, 3, 1, 2)
    tt = 2.0 * np.einsum("jc,kb->kcjb", t1, t1)
    assert np.array_equal(T, T_ref) and np.array_equal(Tp, Tp_ref)
    assert np.abs(S - (2 * T_ref - Tp_ref)).max() < 1e-14 and np.abs(Ut - (2 * T_ref - Tp_ref - tt)).max() < 1e-13
    assert np.abs(Tpt - (Tp_ref + tt)).max() < 1e-13 and np.abs(Th - (2 * t2.transpose(0, 1, 3, 2) - t2)).max() < 1e-14
    ZC = rng.standard_normal((o, o, v, v)); ZB = rng.standard_normal((o, v, v, o))
    dZC, dZB, dY = DeviceBuffer.from_numpy(ZC), DeviceBuffer.from_numpy(ZB), DeviceBuffer(v * v)
    check(qlib.qemb_op_ccsd_y_traces(o, v, dZC.ptr, dZB.ptr, dY.ptr))
    Y_ref = 2.0 * np.einsum("kkac->ac", ZC) - np.einsum("kcak->ac", ZB)
    assert np.abs(dY.numpy((v, v)) - Y_ref).max() < 1e-12
    # small-K update: batched A / shared B, shared A / batched B, K longer than one 32-step
    for (batch, M, N, K, sharedA, sharedB) in ((o * o, v, v, o, False, True), (o * v, v, o, o, True, False), (3, 2 * v + 1, v, 37, False, False)):
        A = rng.standard_normal((1 if sharedA else batch, K, M)); B = rng.standard_normal((1 if sharedB else batch, K, N)); C0 = rng.standard_normal((batch, M, N))
        dA, dB, dC = DeviceBuffer.from_numpy(A), DeviceBuffer.from_numpy(B), DeviceBuffer.from_numpy(C0)
        check(qlib.qemb_op_small_k_update(batch, M, N, K, -0.7, dA.ptr, 0 if sharedA else K * M, dB.ptr, 0 if sharedB else K * N, dC.ptr, M * N))
        ref = C0 - 0.7 * np.einsum("zkm,zkn->zmn", np.broadcast_to(A, (batch, K, M)), np.broadcast_to(B, (batch, K, N)))
        assert np.abs(dC.numpy((batch, M, N)) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_cu_partitioned_contexts_run_kernels(qlib):
    """qemb_ctx_partition(2): contexts created afterwards live on interleaved halves of the compute units (CU-masked streams).  Two host threads, one per new
    context, run a GEMM large enough to need several rounds of workgroups on half the chip; results equal NumPy and each other bit for bit."""
    import threading
    have = qlib.qemb_ctx_count(1)
    assert have >= 1
    check(qlib.qemb_ctx_partition(2))
    try:
        n = qlib.qemb_ctx_count(have + 2)
        assert n >= have + 2
        rng = np.random.default_rng(3)
        A = rng.standard_normal((1, 700, 300)); B = rng.standard_normal((1, 300, 900)); C0 = np.zeros((1, 700, 900))
        out, err = {}, []

        def work(k):
            try:
                check(qlib.qemb_ctx_bind(k))
                out[k] = _gemm(qlib, A, B, C0, 1.0, 0.0, 1, 0)
            except Exception as e:  # noqa: BLE001
                err.append(e)
        ts = [threading.Thread(target=work, args=(have + i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not err, err
        ref = A @ B
        assert np.abs(out[have] - ref).max() < 1e-11 * np.abs(ref).max() and np.array_equal(out[have], out[have + 1])
        assert qlib.qemb_ctx_partition(9) < 0
        check(qlib.qemb_ctx_partition(0))      # existing contexts are drained and get plain streams again; they keep working
        out2 = {}
        t = threading.Thread(target=lambda: (check(qlib.qemb_ctx_bind(have)), out2.update(r=_gemm(qlib, A, B, C0, 1.0, 0.0, 1, 0))))
        t.start(); t.join()
        assert np.array_equal(out2["r"], out[have])
    finally:
        check(qlib.qemb_ctx_partition(0))


@pytest.mark.parametrize("shape", [(21, 21, 9261), (1, 32, 1024), (32, 32, 5000), (7, 3, 2049), (20, 31, 40000)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_tiny_result_long_k(qlib, shape, a_kc, b_kc):
    """Tiny results with a long K (the shapes of the Fvv' / T1 products of small fragments; split-K of a single tile): every operand layout, accumulate mode, odd
    sizes.  (The skinny slab kernel those callers opt into is exercised by the small-fragment solves of test_gpu_fragment.py against the oracle.)"""
    M, N, K = shape
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    got = _gemm(qlib, A, B, C0, -0.7, 0.3, a_kc, b_kc)
    ref = -0.7 * (A @ B) + 0.3 * C0
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("o,v", [(1, 1), (2, 3), (4, 33), (7, 70), (21, 21), (5, 64), (3, 97)])
def test_ccsd_update_fused_passes(qlib, o, v):
    """The fused passes of update_amps against NumPy: Woooo packed from its four terms, the four small T1 products, the double matrix-vector pass,
    the finishing pass that takes the ring products where the GEMMs leave them, the transposing pass with a second output, and the addends of
    scatter_pm_rows / y_traces."""
    rng = np.random.default_rng(31 * o + v)
    nov = o * v
    # --- pack_w_pm_sum == pack_w_pm of the summed tensor
    Wp, X, O1 = (rng.standard_normal((o, o, o, o)) for _ in range(3))
    W = Wp.transpose(2, 3, 0, 1) + X.transpose(2, 3, 0, 1) + O1.transpose(2, 3, 1, 0) + O1.transpose(3, 2, 0, 1)      # W[k,l,i,j] = Wt[ijkl] + X[ijkl] + At[jikl] + At[ijlk]
    npo, nmo = o * (o + 1) // 2, o * (o - 1) // 2
    lwp, lwm = npo + (npo & 1), max(2, nmo + (nmo & 1))
    bufs = [DeviceBuffer.from_numpy(a) for a in (Wp, X, O1, W)]
    outs = [DeviceBuffer(npo * lwp), DeviceBuffer(max(nmo, 1) * lwm), DeviceBuffer(npo * lwp), DeviceBuffer(max(nmo, 1) * lwm)]
    check(qlib.qemb_op_pack_w_pm_sum(o, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, outs[0].ptr, lwp, outs[1].ptr, lwm))
    check(qlib.qemb_op_pack_w_pm(o, bufs[3].ptr, outs[2].ptr, lwp, outs[3].ptr, lwm))
    assert np.abs(outs[0].numpy((npo, lwp)) - outs[2].numpy((npo, lwp))).max() < 1e-13
    if nmo:
        assert np.abs(outs[1].numpy((nmo, lwm)) - outs[3].numpy((nmo, lwm))).max() < 1e-13
    # --- t1_small
    t1, Fov = rng.standard_normal((o, v)), rng.standard_normal((o, v))
    Lvv, Loo = rng.standard_normal((v, v)), rng.standard_normal((o, o))
    d = [DeviceBuffer.from_numpy(a) for a in (t1, Lvv, Loo, Fov)]
    dt1n = DeviceBuffer(nov)
    check(qlib.qemb_op_ccsd_t1_small(o, v, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, dt1n.ptr))
    ref = t1 @ Lvv.T - Loo.T @ t1 + (t1 @ Fov.T) @ t1
    assert np.abs(dt1n.numpy((o, v)) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # --- gemv_rows2
    T1, T2, x1, x2, y0 = rng.standard_normal((nov, nov)), rng.standard_normal((nov, nov)), rng.standard_normal(nov), rng.standard_normal(nov), rng.standard_normal(nov)
    d = [DeviceBuffer.from_numpy(a) for a in (T1, x1, T2, x2, y0)]
    check(qlib.qemb_op_gemv_rows2(nov, nov, d[0].ptr, nov, d[1].ptr, d[2].ptr, nov, d[3].ptr, d[4].ptr, 0.75, 1.0))
    ref = 0.75 * (T1 @ x1 + T2 @ x2) + y0
    assert np.abs(d[4].numpy((nov,)) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # --- the whole T1 right-hand side in one launch: small products + the two long rows + slab sums
    Sm, Lph1 = rng.standard_normal((nov, nov)), rng.standard_normal((nov, nov))
    SA, SB = 5, 3
    PA, PB = rng.standard_normal((SA, nov)), rng.standard_normal((SB, nov))
    d2 = [DeviceBuffer.from_numpy(a) for a in (t1, Lvv, Loo, Fov, Sm, Lph1, PA, PB)]
    check(qlib.qemb_op_ccsd_t1_assemble(o, v, d2[0].ptr, d2[1].ptr, d2[2].ptr, d2[3].ptr, d2[4].ptr, d2[5].ptr, d2[6].ptr, SA, nov, d2[7].ptr, SB, nov, dt1n.ptr))
    ref = (t1 @ Lvv.T - Loo.T @ t1 + (t1 @ Fov.T) @ t1).ravel() + Sm @ Fov.ravel() + Lph1 @ t1.ravel() + PA.sum(0) - PB.sum(0)
    assert np.abs(dt1n.numpy((nov,)) - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    # --- two independent matrix-vector passes in one launch
    Ta, Tb = rng.standard_normal((nov, nov)), rng.standard_normal((o * o, nov))
    xa, ya, yb = rng.standard_normal(nov), rng.standard_normal(nov), rng.standard_normal(o * o)
    d2 = [DeviceBuffer.from_numpy(a) for a in (Ta, xa, ya, Tb, yb)]
    check(qlib.qemb_op_gemv_rows_two(nov, nov, d2[0].ptr, nov, d2[1].ptr, d2[2].ptr, 1.0, 0.0, o * o, nov, d2[3].ptr, nov, d2[1].ptr, d2[4].ptr, 0.5, 1.0))
    assert np.abs(d2[2].numpy((nov,)) - Ta @ xa).max() < 1e-12 * max(1.0, np.abs(Ta @ xa).max())
    assert np.abs(d2[4].numpy((o * o,)) - (0.5 * (Tb @ xa) + yb)).max() < 1e-12 * max(1.0, np.abs(Tb @ xa).max())
    # --- finish_t2_rings (t2n and OV symmetric under (ij)(ab), as the ladder and the integrals leave them)
    def sym(x):
        return 0.5 * (x + x.transpose(1, 0, 3, 2))
    t2n, OV = sym(rng.standard_normal((o, o, v, v))), sym(rng.standard_normal((o, o, v, v)))
    U, RS, M = rng.standard_normal((o, o, v, v)), rng.standard_normal((o, v, o, v)), rng.standard_normal((o, v, o, v))
    eo, ev = -1.0 - rng.random(o), 1.0 + rng.random(v)
    t1n = rng.standard_normal((o, v))
    F = U + RS.transpose(0, 2, 1, 3) - 0.5 * M.transpose(0, 2, 1, 3) - M.transpose(0, 2, 3, 1)
    D = eo[:, None, None, None] + eo[None, :, None, None] - ev[None, None, :, None] - ev[None, None, None, :]
    ref = (t2n + OV + F + F.transpose(1, 0, 3, 2)) / D
    d = [DeviceBuffer.from_numpy(a) for a in (t2n, U, OV, RS, M, eo, ev, t1n)]
    check(qlib.qemb_op_ccsd_finish_t2_rings(o, v, *[b.ptr for b in d]))
    got = d[0].numpy((o, o, v, v))
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    gt = got.transpose(1, 0, 3, 2)
    assert all(np.array_equal(got[i, j], gt[i, j]) for i in range(o) for j in range(o) if i != j)      # pairs of tiles are stored both ways from one result
    assert np.abs(got - gt).max() < 1e-13 * max(1.0, np.abs(ref).max())
    assert np.abs(d[7].numpy((o, v)) - t1n / (eo[:, None] - ev[None, :])).max() < 1e-13
    # --- transposing pass with base and second output: W2 = base + ZC[k,i,a,c] at [i,a,k,c],  R = W1 - W2 / 2
    ZC, base, W1 = rng.standard_normal((o, o, v, v)), rng.standard_normal((o, v, o, v)), rng.standard_normal((o, v, o, v))
    d = [DeviceBuffer.from_numpy(a) for a in (ZC, base, W1)]
    dW2, dR = DeviceBuffer(o * o * v * v), DeviceBuffer(o * o * v * v)
    perm = (1, 2, 0, 3)
    out_shape = tuple(ZC.shape[p] for p in perm)
    st = [int(np.prod(out_shape[k + 1:])) for k in range(4)]
    ostr = [0] * 4
    for k, p in enumerate(perm):
        ostr[p] = st[k]
    istr = [int(np.prod(ZC.shape[k + 1:])) for k in range(4)]
    check(qlib.qemb_op_copy4_two(i64x4(ZC.shape), d[0].ptr, i64x4(istr), dW2.ptr, i64x4(ostr), 1.0, 1.0, d[1].ptr, dR.ptr, d[2].ptr, 1.0, -0.5))
    W2_ref = base + ZC.transpose(perm)
    assert np.abs(dW2.numpy(out_shape) - W2_ref).max() < 1e-14 and np.abs(dR.numpy(out_shape) - (W1 - 0.5 * W2_ref)).max() < 1e-14
    # --- addends
    ncols = 3 * v + 1
    Xp, Xm, add = rng.standard_normal((npo, ncols)), rng.standard_normal((max(nmo, 1), ncols)), rng.standard_normal((o, o, ncols))
    d = [DeviceBuffer.from_numpy(a) for a in (Xp, Xm, add)]
    dO = DeviceBuffer(o * o * ncols)
    check(qlib.qemb_op_scatter_pm_rows_add(o, ncols, d[0].ptr, d[1].ptr, dO.ptr, d[2].ptr))
    ref = add.copy()
    for i in range(o):
        for j in range(i + 1):
            p = Xp[i * (i + 1) // 2 + j]
            if i > j:
                m = Xm[i * (i - 1) // 2 + j]
                ref[i, j] += p + m; ref[j, i] += p - m
            else:
                ref[i, j] += p
    assert np.abs(dO.numpy((o, o, ncols)) - ref).max() < 1e-14
    ZCt, ZBt, addv = rng.standard_normal((o, o, v, v)), rng.standard_normal((o, v, v, o)), rng.standard_normal((v, v))
    d = [DeviceBuffer.from_numpy(a) for a in (ZCt, ZBt, addv)]
    dY = DeviceBuffer(v * v)
    check(qlib.qemb_op_ccsd_y_traces_add(o, v, d[0].ptr, d[1].ptr, dY.ptr, d[2].ptr))
    assert np.abs(dY.numpy((v, v)) - (2.0 * np.einsum("kkac->ac", ZCt) - np.einsum("kcak->ac", ZBt) + addv)).max() < 1e-12
    slabs = rng.standard_normal((7, v, v))
    dS = DeviceBuffer.from_numpy(slabs)
    check(qlib.qemb_op_ccsd_y_traces_slabs(o, v, d[0].ptr, d[1].ptr, dY.ptr, dS.ptr, 7, v * v, -1.0))
    assert np.abs(dY.numpy((v, v)) - (2.0 * np.einsum("kkac->ac", ZCt) - np.einsum("kcak->ac", ZBt) - slabs.sum(0))).max() < 1e-12


@pytest.mark.parametrize("o,v,m", [(1, 1, 1), (3, 5, 2), (4, 33, 6), (7, 70, 8), (21, 21, 6), (20, 100, 6)])
def test_ccsd_iteration_end_single_launch_kernels(qlib, o, v, m):
    """diis_push (error vector, stored trial vector, Gram row) and ccsd_extrapolate_energy (amplitudes, tau, energy) -- one launch each, the last
    workgroup finishing the reduction -- against NumPy; launched repeatedly so that the workgroup counter is seen to be left at zero."""
    rng = np.random.default_rng(7 * o + v + m)
    nov, n = o * v, o * v + o * o * v * v
    trial, prev = rng.standard_normal(n), rng.standard_normal(n)
    es = [rng.standard_normal(n) for _ in range(m)]
    for self_slot in sorted({0, m - 1, m // 2}):
        dtrial, dprev = DeviceBuffer.from_numpy(trial), DeviceBuffer.from_numpy(prev)
        des = [DeviceBuffer.from_numpy(e) for e in es]
        dx, drow = DeviceBuffer(n), DeviceBuffer(8)
        ys = (C.c_void_p * m)(*[b.ptr for b in des])
        row = (C.c_double * 8)()
        for rep in range(3):
            check(qlib.qemb_op_diis_push(n, dtrial.ptr, dprev.ptr, des[self_slot].ptr, dx.ptr, m, ys, self_slot, drow.ptr, row))
            e_ref = trial - prev
            ref = np.array([e_ref @ (e_ref if j == self_slot else es[j]) for j in range(m)])
            got = np.array(row[:m])
            assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), (rep, got, ref)
            assert np.array_equal(drow.numpy((8,))[:m], got)
        assert np.array_equal(des[self_slot].numpy((n,)), e_ref) and np.array_equal(dx.numpy((n,)), trial)
        # in place: the stored copy of the trial vector aliases the previous vector (the first, DIIS-less iteration)
        check(qlib.qemb_op_diis_push(n, dtrial.ptr, dprev.ptr, des[self_slot].ptr, dprev.ptr, 1, (C.c_void_p * 1)(des[self_slot].ptr), 0, drow.ptr, row))
        assert np.array_equal(dprev.numpy((n,)), trial) and abs(row[0] - e_ref @ e_ref) < 1e-11 * max(1.0, e_ref @ e_ref)
        for b in [dtrial, dprev, dx, drow] + des:
            b.free()
    xs = [rng.standard_normal(n) for _ in range(m)]
    coef = rng.standard_normal(m)
    L = rng.standard_normal(o * o * v * v)
    dxs = [DeviceBuffer.from_numpy(x) for x in xs]
    damp, dL, dtau = DeviceBuffer(n), DeviceBuffer.from_numpy(L), DeviceBuffer(o * o * v * v)
    e = C.c_double()
    amp_ref = sum(c * x for c, x in zip(coef, xs))
    t1, t2 = amp_ref[:nov].reshape(o, v), amp_ref[nov:].reshape(o, o, v, v)
    tau_ref = t2 + np.einsum("ia,jb->ijab", t1, t1)
    for rep in range(3):
        check(qlib.qemb_op_ccsd_extrapolate_energy(o, v, m, (C.c_double * m)(*coef), (C.c_void_p * m)(*[b.ptr for b in dxs]), damp.ptr, dL.ptr, dtau.ptr, C.byref(e)))
        assert np.abs(damp.numpy((n,)) - amp_ref).max() < 1e-13 * max(1.0, np.abs(amp_ref).max())
        assert np.abs(dtau.numpy((o, o, v, v)) - tau_ref).max() < 1e-12 * max(1.0, np.abs(tau_ref).max())
        assert abs(e.value - L @ tau_ref.ravel()) < 1e-10 * max(1.0, np.abs(L).sum() ** 0.5 * np.abs(tau_ref).max())
    # nothing to extrapolate, amplitudes already in place: only tau and the energy
    amp0 = damp.numpy((n,)).copy()
    check(qlib.qemb_op_ccsd_extrapolate_energy(o, v, 1, (C.c_double * 1)(1.0), (C.c_void_p * 1)(damp.ptr), damp.ptr, dL.ptr, dtau.ptr, C.byref(e)))
    assert np.array_equal(damp.numpy((n,)), amp0)
    t1, t2 = amp0[:nov].reshape(o, v), amp0[nov:].reshape(o, o, v, v)
    assert abs(e.value - L @ (t2 + np.einsum("ia,jb->ijab", t1, t1)).ravel()) < 1e-10 * max(1.0, np.abs(L).sum() ** 0.5 * np.abs(tau_ref).max())


@pytest.mark.parametrize("shape", [(37, 200, 200, 20, False, True), (5, 70, 64, 32, True, False), (4, 33, 129, 7, False, False), (3, 200, 256, 20, False, True),
                                   (2, 100, 257, 20, False, True), (6, 31, 63, 5, False, False), (400, 200, 200, 20, False, True)])
def test_small_k_update_strip_and_tile_variants(qlib, shape):
    """C[z] += alpha A[z]^T B[z] with K <= 32: the 32-row strip kernel (64 <= N <= 256) and the 32 x 32 tile kernel on both sides of its
    limits; the last shape is the rank-n_occ update of the o^2 v^2 tensor at the benchmark size."""
    batch, M, N, K, sharedA, sharedB = shape
    rng = np.random.default_rng(batch + 3 * M + 5 * N + 7 * K)
    A = rng.standard_normal((1 if sharedA else batch, K, M)); B = rng.standard_normal((1 if sharedB else batch, K, N)); C0 = rng.standard_normal((batch, M, N))
    dA, dB, dC = DeviceBuffer.from_numpy(A), DeviceBuffer.from_numpy(B), DeviceBuffer.from_numpy(C0)
    check(qlib.qemb_op_small_k_update(batch, M, N, K, 1.3, dA.ptr, 0 if sharedA else K * M, dB.ptr, 0 if sharedB else K * N, dC.ptr, M * N))
    ref = C0 + 1.3 * np.einsum("zkm,zkn->zmn", np.broadcast_to(A, (batch, K, M)), np.broadcast_to(B, (batch, K, N)))
    assert np.abs(dC.numpy((batch, M, N)) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("o,v", [(1, 3), (2, 1), (4, 5), (7, 40)])
def test_hole_hole_ladder_through_packed_pairs(qlib, o, v):
    """R[ij,ab] = W[klij] tau[klab] (W symmetric under (k,l,i,j) -> (l,k,j,i)) from the (+/-) packed images of W, the packed tau rows as the
    right operand and the two-pair scatter: == the dense contraction; assign and accumulate modes."""
    rng = np.random.default_rng(10 * o + v)
    W = rng.standard_normal((o, o, o, o)); W = W + W.transpose(1, 0, 3, 2)
    tau = rng.standard_normal((o, o, v, v)); tau = tau + tau.transpose(1, 0, 3, 2)
    npo, nmo, npv, nmv = o * (o + 1) // 2, o * (o - 1) // 2, v * (v + 1) // 2, v * (v - 1) // 2
    ldp, ldm = npv + (npv & 1), max(nmv + (nmv & 1), 2)
    lwp, lwm = npo + (npo & 1), max(nmo + (nmo & 1), 2)
    dW, dtau = DeviceBuffer.from_numpy(W), DeviceBuffer.from_numpy(tau)
    dAp, dAm = DeviceBuffer(npo * lwp), DeviceBuffer(max(nmo, 1) * lwm)
    check(qlib.qemb_op_pack_w_pm(o, dW.ptr, dAp.ptr, lwp, dAm.ptr, lwm))
    dTp, dTm = DeviceBuffer(npo * ldp), DeviceBuffer(max(nmo, 1) * ldm)
    check(qlib.qemb_op_ladder_pack_tau(o, v, dtau.ptr, dTp.ptr, ldp, dTm.ptr, ldm))
    dHp, dHm = DeviceBuffer(npo * ldp), DeviceBuffer(max(nmo, 1) * ldm)
    check(qlib.qemb_op_gemm(npo, npv, npo, 1.0, dAp.ptr, lwp, 1, 0, dTp.ptr, ldp, 0, 0, 0.0, dHp.ptr, ldp, 0, 1))
    if nmo and nmv:
        check(qlib.qemb_op_gemm(nmo, nmv, nmo, 1.0, dAm.ptr, lwm, 1, 0, dTm.ptr, ldm, 0, 0, 0.0, dHm.ptr, ldm, 0, 1))
    zero_p, zero_m = DeviceBuffer.from_numpy(np.zeros(npo * ldp)), DeviceBuffer.from_numpy(np.zeros(max(nmo, 1) * ldm))
    ref = np.einsum("klij,klab->ijab", W, tau)
    base = rng.standard_normal((o, o, v, v))
    d2 = DeviceBuffer.from_numpy(base)
    check(qlib.qemb_op_ladder_scatter_pm2(o, v, zero_p.ptr, ldp, zero_m.ptr, ldm, dHp.ptr, dHm.ptr if (nmo and nmv) else None, 1, d2.ptr))
    assert np.abs(d2.numpy(ref.shape) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())          # assign: the old content is gone
    d3 = DeviceBuffer.from_numpy(base)
    check(qlib.qemb_op_ladder_scatter_pm2(o, v, zero_p.ptr, ldp, zero_m.ptr, ldm, dHp.ptr, dHm.ptr if (nmo and nmv) else None, 0, d3.ptr))
    assert np.abs(d3.numpy(ref.shape) - (base + ref)).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_gather_and_scale_rows(qlib):
    rng = np.random.default_rng(12)
    src = rng.standard_normal((9, 13))                       # rows of 13 with ld 13; gather 11 columns of each
    idx = np.array([3, -1, 0, 8, 8, -1, 5], dtype=np.int64)
    dsrc, ddst = DeviceBuffer.from_numpy(src), DeviceBuffer(len(idx) * 11)
    didx = DeviceBuffer(len(idx)); check(qlib.qemb_h2d(didx.ptr, idx.ctypes.data, idx.nbytes))
    check(qlib.qemb_op_gather_rows(len(idx), 11, didx.ptr, dsrc.ptr, 13, ddst.ptr))
    ref = np.where(idx[:, None] >= 0, src[np.maximum(idx, 0), :11], 0.0)
    assert np.array_equal(ddst.numpy((len(idx), 11)), ref)
    x = rng.standard_normal((6, 300)); sc = np.array([1.0, 0.0, 2.5, 1.0, -1.0, 0.0])
    x[1, 7] = np.inf                                          # a row scaled by zero is cleared, not multiplied
    dx, ds = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(sc)
    check(qlib.qemb_op_scale_rows(6, 300, dx.ptr, ds.ptr))
    with np.errstate(invalid="ignore"):
        ref = x * sc[:, None]
    ref[1] = 0.0; ref[5] = 0.0
    assert np.array_equal(dx.numpy((6, 300)), ref)


# ---- the tile configurations the headline benchmark runs on (BASELINE configs[2]: o = 20, v = 200) -----------------------------
# cfg 13 / 23: 224 x 128 tile as 2 x 4 waves (7 x 2 MFMA tiles per wave) -- the (+) pair block of the pp-ladder (M = npair(20) = 210)
#              and the K = n products of the MO transformation when 192 < n <= 224;
# cfg 15 / 25: 192 x 128 tile (6 x 2 per wave) -- the (-) pair block (M = 190);
# cfg 20 / 21: 128 x 32 / 32 x 128 tiles -- the products with an n_occ-sized side (ccsd.cpp:249).
# 23 / 25 are 13 / 15 under the ladder's own kernel symbol (ccsd.cpp:217), i.e. separately compiled instantiations.
@pytest.mark.parametrize("cfg,M", [(13, 210), (23, 210), (13, 224), (13, 220), (15, 190), (25, 190), (15, 192), (23, 97), (213, 210), (215, 190), (33, 220), (33, 112), (33, 113), (35, 160), (35, 465), (35, 161), (35, 153), (35, 136), (4, 120), (4, 128), (11, 105), (12, 45), (36, 78), (36, 66), (36, 80), (36, 81), (37, 96), (37, 100), (37, 190)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("ks", [0, 8])
def test_gemm_ladder_tile_configs(qlib, cfg, M, a_kc, b_kc, ks):
    """bench-shaped products: one m-tile holding every packed pair row, N >= 2048 columns, long K (with a k-tail), split-K 0 / 8."""
    rng = np.random.default_rng(1000 * cfg + M + 2 * a_kc + b_kc)
    N, K = 2304 + 2 * (M % 7), 4100
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    qlib.qemb_set_gemm_ksplit(ks)
    try:
        got = _gemm(qlib, A, B, C0, 0.75, -0.5, a_kc, b_kc, cfg=cfg)
        again = _gemm(qlib, A, B, C0, 0.75, -0.5, a_kc, b_kc, cfg=cfg)
    finally:
        qlib.qemb_set_gemm_ksplit(0)
    ref = 0.75 * (A @ B) + -0.5 * C0
    assert np.abs(got - ref).max() < 1e-12 * K, (cfg, M, a_kc, b_kc, ks, np.abs(got - ref).max())
    assert np.array_equal(got, again), "split-K slab reduction must be run-to-run deterministic"


@pytest.mark.parametrize("cfg,shape", [(13, (210, 2050, 1023)), (15, (190, 2049, 515)), (23, (210, 2049, 4097)), (25, (189, 2051, 4099))])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (0, 0)])
def test_gemm_ladder_tiles_scalar_load_variant(qlib, cfg, shape, a_kc, b_kc):
    """odd extents / leading dimensions: the VEC = 1 instantiations of the same tiles (guarded scalar loads throughout)."""
    M, N, K = shape
    rng = np.random.default_rng(cfg * 31 + M + N + K)
    A = rng.standard_normal((2, M, K)); B = rng.standard_normal((2, K, N)); C0 = rng.standard_normal((2, M, N))
    got = _gemm(qlib, A, B, C0, 1.25, 0.5, a_kc, b_kc, cfg=cfg)
    ref = 1.25 * np.einsum("bmk,bkn->bmn", A, B) + 0.5 * C0
    assert np.abs(got - ref).max() < 1e-12 * K


@pytest.mark.parametrize("cfg,shape", [(20, (8000, 20, 200)), (20, (4001, 32, 200)), (20, (2000, 20, 20000)), (20, (300, 7, 51)),
                                       (21, (20, 8000, 200)), (21, (32, 4001, 200)), (21, (20, 200, 84000)), (21, (7, 300, 51))])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_skinny_tile_configs(qlib, cfg, shape, a_kc, b_kc):
    """128 x 32 / 32 x 128 tiles of the t1 contractions (N or M = n_occ <= 32); the K = o v^2-long ones also split K."""
    M, N, K = shape
    rng = np.random.default_rng(cfg + M + 3 * N + 5 * K)
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    got = _gemm(qlib, A, B, C0, -1.0, 1.0, a_kc, b_kc, cfg=cfg)
    ref = -(A @ B) + C0
    assert np.abs(got - ref).max() < 1e-12 * max(K, 100)


def test_gemm_skinny_batched_as_in_update_amps(qlib):
    """the batched 32 x 128 call of ccsd.cpp (ZC[k,i,a,c] = t1[i,d] ovvv[k,d,a,c]): batch = o, shared A (stride 0)."""
    o, v = 20, 64
    rng = np.random.default_rng(3)
    t1 = rng.standard_normal((o, v)); ovvv = rng.standard_normal((o, v, v * v))
    dA, dB, dC = DeviceBuffer.from_numpy(t1), DeviceBuffer.from_numpy(ovvv), DeviceBuffer(o * o * v * v)
    qlib.qemb_set_gemm_config(21)
    try:
        check(qlib.qemb_op_gemm(o, v * v, v, 1.0, dA.ptr, v, 1, 0, dB.ptr, v * v, 0, v * v * v, 0.0, dC.ptr, v * v, o * v * v, o))
    finally:
        qlib.qemb_set_gemm_config(-1)
    ref = np.einsum("id,kdx->kix", t1, ovvv)
    assert np.abs(dC.numpy((o, o, v * v)) - ref).max() < 1e-11


@pytest.mark.parametrize("n", [220, 224, 200, 193])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0)])
def test_gemm_tall_products_on_the_128x224_tile(qlib, n, a_kc, b_kc):
    """cfg 34: M long, 192 < N <= 224 columns in ONE tile (the flat slab . C product of mo_transform), K = n with a k-tail."""
    rng = np.random.default_rng(34 + n + a_kc + 2 * b_kc)
    M = 5 * 128 + 77
    A = rng.standard_normal((1, M, n)); B = rng.standard_normal((1, n, n)); C0 = rng.standard_normal((1, M, n))
    got = _gemm(qlib, A, B, C0, 1.0, 0.0, a_kc, b_kc, cfg=34)
    assert np.abs(got - A @ B).max() < 1e-11 * n
    got = _gemm(qlib, A, B, C0, -0.5, 2.0, a_kc, b_kc, cfg=34)
    assert np.abs(got - (-0.5 * (A @ B) + 2.0 * C0)).max() < 1e-11 * n


@pytest.mark.parametrize("n,cfg,big", [(220, 13, True), (200, 13, False), (193, 13, False), (201, 4, True), (220, 33, True), (222, 33, False)])
def test_gemm_mo_transform_products_at_bench_size(qlib, n, cfg, big):
    """the two product shapes of mo_transform (ccsd.cpp:49-65) at 192 < n <= 224 with the 224 x 128 tile: the TN quarter transform
    Out[x',(rest)] = C[x,x'] In[(rest),x] (M = K = n, N long) and the batched slab products (M = N = K = n, batch = pairs)."""
    rng = np.random.default_rng(n)
    ncol = 3000 + (n & 1)
    Cm = rng.standard_normal((n, n)); X = rng.standard_normal((ncol, n))
    dC, dX, dO = DeviceBuffer.from_numpy(Cm), DeviceBuffer.from_numpy(X), DeviceBuffer(n * ncol)
    qlib.qemb_set_gemm_config(cfg)
    try:
        # A(m,k) = C[k*n + m] (not k-contiguous), B(k,nn) = X[nn*n + k] (k-contiguous)
        check(qlib.qemb_op_gemm(n, ncol, n, 1.0, dC.ptr, n, 0, 0, dX.ptr, n, 1, 0, 0.0, dO.ptr, ncol, 0, 1))
        assert np.abs(dO.numpy((n, ncol)) - Cm.T @ X.T).max() < 1e-11 * n
        nb = 520 if big else 37
        S = rng.standard_normal((nb, n, n))
        dS, dR = DeviceBuffer.from_numpy(S), DeviceBuffer(nb * n * n)
        check(qlib.qemb_op_gemm(n, n, n, 1.0, dS.ptr, n, 1, n * n, dC.ptr, n, 0, 0, 0.0, dR.ptr, n, n * n, nb))     # slab . C
        assert np.abs(dR.numpy((nb, n, n)) - S @ Cm).max() < 1e-11 * n
        check(qlib.qemb_op_gemm(n, n, n, 1.0, dC.ptr, n, 0, 0, dS.ptr, n, 0, n * n, 0.0, dR.ptr, n, n * n, nb))     # C^T . slab
        assert np.abs(dR.numpy((nb, n, n)) - Cm.T @ S).max() < 1e-11 * n
    finally:
        qlib.qemb_set_gemm_config(-1)


def test_pm_packed_ladder_at_bench_tiles(qlib):
    """o = 20 with v = 66: npair(o) = 210 rows select the 224 x 128 tile (cfg 23), the 190 antisymmetric rows the 192 x 128 tile
    (cfg 25), npair(v) = 2211 >= 2048 columns -- the dispatch of CcsdSolver::apply_ladder at the benchmark's o -- against the
    dense sum_cd (ac|bd) tau_ijcd."""
    o, v = 20, 66
    rng = np.random.default_rng(o * 100 + v)
    npv, nmv, npo, nmo = v * (v + 1) // 2, v * (v - 1) // 2, o * (o + 1) // 2, o * (o - 1) // 2
    ldp, ldm = npv + (npv & 1), nmv + (nmv & 1)
    Bv = rng.standard_normal((40, v, v)); Bv = Bv + Bv.transpose(0, 2, 1)
    vv = np.einsum("Pac,Pbd->acbd", Bv, Bv, optimize=True)                   # (ac|bd) at [a,c,b,d]
    tau = rng.standard_normal((o, o, v, v)); tau = tau + tau.transpose(1, 0, 3, 2)
    il, sl = np.tril_indices(v), np.tril_indices(v, -1)
    Vac = vv.transpose(0, 2, 1, 3)                                           # [a,b,c,d] = (ac|bd)
    Vp = np.zeros((npv, ldp)); Vm = np.zeros((max(nmv, 1), ldm))
    Vp[:, :npv] = (Vac + Vac.transpose(0, 1, 3, 2))[il[0], il[1]][:, il[0], il[1]]
    Vm[:, :nmv] = (Vac - Vac.transpose(0, 1, 3, 2))[sl[0], sl[1]][:, sl[0], sl[1]]
    dT = DeviceBuffer.from_numpy(tau)
    dVp, dVm = DeviceBuffer.from_numpy(Vp), DeviceBuffer.from_numpy(Vm)
    dTp, dTm = DeviceBuffer(npo * ldp), DeviceBuffer(nmo * ldm)
    dRp, dRm = DeviceBuffer(npo * ldp), DeviceBuffer(nmo * ldm)
    check(qlib.qemb_op_ladder_pack_tau(o, v, dT.ptr, dTp.ptr, ldp, dTm.ptr, ldm))
    for cfg_p, cfg_m, ks in ((23, 25, 0), (23, 25, 8), (13, 15, 3)):
        qlib.qemb_set_gemm_ksplit(ks)
        try:
            qlib.qemb_set_gemm_config(cfg_p)
            check(qlib.qemb_op_gemm(npo, npv, ldp, 1.0, dTp.ptr, ldp, 1, 0, dVp.ptr, ldp, 1, 0, 0.0, dRp.ptr, ldp, 0, 1))
            qlib.qemb_set_gemm_config(cfg_m)
            check(qlib.qemb_op_gemm(nmo, nmv, ldm, 1.0, dTm.ptr, ldm, 1, 0, dVm.ptr, ldm, 1, 0, 0.0, dRm.ptr, ldm, 0, 1))
        finally:
            qlib.qemb_set_gemm_config(-1); qlib.qemb_set_gemm_ksplit(0)
        t2 = rng.standard_normal((o, o, v, v))
        d2 = DeviceBuffer.from_numpy(t2)
        check(qlib.qemb_op_ladder_scatter_pm(o, v, dRp.ptr, ldp, dRm.ptr, ldm, d2.ptr))
        ref = t2 + np.einsum("abcd,ijcd->ijab", Vac, tau, optimize=True)
        assert np.abs(d2.numpy(t2.shape) - ref).max() < 1e-10 * np.abs(ref).max(), (cfg_p, cfg_m, ks)


def test_device_timers_hold_a_bounded_number_of_events(qlib):
    """A long run that never reads its timers must not accumulate HIP events (dev_ops_hip.hip: laps are harvested and their event
    pairs recycled), and a lap that was begun but never ended (early return of the bracketed region) must not poison the slot."""
    slot = 9
    check(qlib.qemb_timer_reset(slot))
    x = DeviceBuffer.from_numpy(np.ones(1 << 16)); out = DeviceBuffer(2)
    for _ in range(1000):
        check(qlib.qemb_timer_begin(slot))
        check(qlib.qemb_op_dot(1 << 16, x.ptr, x.ptr, out.ptr))
        check(qlib.qemb_timer_end(slot))
    assert qlib.qemb_timer_live_events(slot) <= 64
    check(qlib.qemb_timer_begin(slot))            # begun, never ended ...
    check(qlib.qemb_timer_begin(slot))            # ... the next lap reuses the pair
    check(qlib.qemb_op_dot(1 << 16, x.ptr, x.ptr, out.ptr))
    check(qlib.qemb_timer_end(slot))
    check(qlib.qemb_timer_begin(slot))            # and a dangling one at read time is dropped
    ms, cnt = C.c_double(), C.c_int64()
    check(qlib.qemb_timer_read(slot, C.byref(ms), C.byref(cnt)))
    assert cnt.value == 1001 and ms.value > 0.0
    assert qlib.qemb_timer_end(slot) != 0         # nothing open any more
    check(qlib.qemb_timer_reset(slot))


# every tile configuration that runs the MODE 1 main loop in production (inline-asm ds_read_b64 + hand-counted s_waitcnt) against the
# classic loop of the same tile (cfg + 200; 23 / 25 are 13 / 15 under the ladder's kernel symbol): same summation order, so the results
# must be IDENTICAL -- a compiler that moves a fragment register between the asm read and its wait would show up here first.
@pytest.mark.parametrize("cfg,classic,M,N", [(0, 200, 256, 384), (1, 201, 192, 192), (4, 204, 128, 512), (13, 213, 210, 2304), (23, 213, 210, 2304),
                                             (15, 215, 190, 2304), (25, 215, 190, 2304), (33, 233, 220, 512), (34, 234, 640, 220), (35, 235, 465, 512), (36, 236, 78, 1440), (37, 237, 480, 384)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_mode1_equals_classic_loop(qlib, cfg, classic, M, N, a_kc, b_kc):
    rng = np.random.default_rng(7 * cfg + a_kc + 2 * b_kc)
    K = 1080                                   # 67.5 k-tiles of 16: odd tile count and a k-tail
    A = rng.standard_normal((1, M, K)); B = rng.standard_normal((1, K, N)); C0 = rng.standard_normal((1, M, N))
    for ks in (0, 4):
        qlib.qemb_set_gemm_ksplit(ks)
        try:
            new = _gemm(qlib, A, B, C0, 0.75, -0.5, a_kc, b_kc, cfg=cfg)
            old = _gemm(qlib, A, B, C0, 0.75, -0.5, a_kc, b_kc, cfg=classic)
        finally:
            qlib.qemb_set_gemm_ksplit(0)
        assert np.array_equal(new, old), (cfg, classic, a_kc, b_kc, ks, np.abs(new - old).max())
    assert np.abs(new - (0.75 * (A @ B) - 0.5 * C0)).max() < 1e-12 * K


def test_diagnostic_gemm_configs_are_not_reachable(qlib):
    """3xx (stamps) and 4xx-6xx (ablation, wrong by construction) are for tools/ with QEMB_GEMM_DIAGNOSTICS=1 only"""
    if os.environ.get("QEMB_GEMM_DIAGNOSTICS"):
        pytest.skip("diagnostics enabled in this environment")
    A = np.ones((1, 64, 64)); B = np.ones((1, 64, 64)); C0 = np.zeros((1, 64, 64))
    for cfg in (313, 413, 513, 613, 304, 404):
        with pytest.raises(Exception, match="diagnostic"):
            _gemm(qlib, A, B, C0, 1.0, 0.0, 1, 1, cfg=cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("n,npairs,cfg", [(220, 37, 34), (220, 5, -1), (222, 9, 34), (96, 40, -1), (45, 7, -1), (10, 3, 2), (64, 50, 1)])
def test_gemm_slab_rows_is_the_batched_transposed_product(qlib, n, npairs, cfg):
    """GemmDesc::a_slab (round 5): the rows (pair, q) of a stack of n x n slabs X[pair][k][q] read as ONE tall !a_kcontig operand --
    Out[(pair, q)][p'] = sum_k X[pair][k][q] C[k][p'], i.e. (C^T X[pair])^T slab by slab: the last quarter transform of mo_transform as a flat product."""
    rng = np.random.default_rng(n + npairs)
    X = rng.standard_normal((npairs, n, n)); Cm = rng.standard_normal((n, n))
    dX, dC, dO = DeviceBuffer.from_numpy(X), DeviceBuffer.from_numpy(Cm), DeviceBuffer(npairs * n * n)
    check(qlib.qemb_op_gemm_slab_rows(npairs * n, n, n, dX.ptr, n, n, n * n - n, dC.ptr, n, 0, dO.ptr, n, cfg))
    got = dO.numpy((npairs, n, n))
    ref = np.einsum("Pkq,kp->Pqp", X, Cm)
    assert np.abs(got - ref).max() < 1e-12 * n
