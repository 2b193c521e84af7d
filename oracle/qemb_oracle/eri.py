"""Oracle for rows a3 / a4 / a5 (AO -> embedding ERI transforms) and the packed-pair layouts.  Test infra."""
import numpy as np


def npair(n):
    return n * (n + 1) // 2


def ravel_symmetric(a, b):
    """shared/helper.py:260-276: ij = i(i+1)/2 + j for i >= j."""
    return a * (a + 1) // 2 + b if a > b else b * (b + 1) // 2 + a


def restore_s1(eri, n):
    """ao2mo.restore(1, ...) semantics for s8 (1-D), s4 (npair x npair) or s1 input."""
    eri = np.asarray(eri)
    npr = npair(n)
    il = np.tril_indices(n)
    if eri.ndim == 4:
        return eri
    if eri.size == npr * (npr + 1) // 2:       # s8
        s4 = np.zeros((npr, npr))
        ii = np.tril_indices(npr)
        s4[ii] = eri.ravel()
        s4 = s4 + s4.T - np.diag(np.diag(s4))
        eri = s4
    if eri.shape == (npr, npr):                 # s4
        tmp = np.zeros((npr, n, n))
        tmp[:, il[0], il[1]] = eri
        tmp[:, il[1], il[0]] = eri
        out = np.zeros((n, n, n, n))
        out[il[0], il[1]] = tmp
        out[il[1], il[0]] = tmp
        return out
    if eri.shape == (n * n, n * n):
        return eri.reshape(n, n, n, n)
    raise ValueError("unrecognised ERI layout")


def pack_s4(eri1, n=None):
    """s1 -> 4-fold packed (npair, npair), row ij = i(i+1)/2 + j (the layout of dataset f{I}: mbe.py:1039)."""
    eri1 = np.asarray(eri1)
    n = eri1.shape[0] if n is None else n
    il = np.tril_indices(n)
    return np.ascontiguousarray(eri1[il[0], il[1]][:, il[0], il[1]])


def pack_s8(eri1):
    s4 = pack_s4(eri1)
    return np.ascontiguousarray(s4[np.tril_indices(s4.shape[0])])


def ao2mo_full(eri, C, compact=True):
    """Semantics of pyscf ao2mo.incore.full(eri, C, compact=True) as called at mbe.py:1038 / solver.py:900:
    (ij|kl) = sum C_mu,i C_nu,j C_ka,k C_la,l (mu nu|ka la); s4-packed when compact."""
    N = C.shape[0]
    e1 = restore_s1(eri, N)
    t = np.einsum("pqrs,pi->iqrs", e1, C, optimize=True)
    t = np.einsum("iqrs,qj->ijrs", t, C, optimize=True)
    t = np.einsum("ijrs,rk->ijks", t, C, optimize=True)
    t = np.einsum("ijks,sl->ijkl", t, C, optimize=True)
    return pack_s4(t) if compact else t


def integral_direct_DF(pqL, j2c, TA):
    """Restates molbe/eri_onthefly.py:108-144 for one fragment given the AO 3-index integrals.

    pqL: (N, N, naux) = (mu nu|P); j2c: (naux, naux) = (P|Q); TA: (N, n).
    low = cholesky(j2c) (:108); Lqi = Lqp @ TA (:134); Lij = Liq @ TA (:136);
    bb = low^-1 b (:141); eri = bb^T bb (:143); restore('4') (:144)."""
    naux = pqL.shape[2]
    n = TA.shape[1]
    low = np.linalg.cholesky(j2c)
    Lqp = np.transpose(pqL, (2, 1, 0))
    Lqi = Lqp @ TA
    Liq = np.moveaxis(Lqi, 2, 1)
    Lij = Liq @ TA
    b = Lij.reshape(naux, -1)
    import scipy.linalg
    bb = scipy.linalg.solve_triangular(low, b, lower=True)
    eri_nosym = bb.T @ bb
    return pack_s4(eri_nosym.reshape(n, n, n, n))


def df_transform_packed(P_munu_packed, L_PQ, TA):
    """Rows a5 maths (dense, unscreened): (P|mu nu) given for mu>=nu pairs as (naux, npair(N)).
    Follows _cpp/eri_sparse_DF.cpp:484-621: two contractions with TA to (P|ij) i<=j, then
    X = L^-1 (P|ij), eri = X^T X in 4-fold packed form."""
    import scipy.linalg
    naux, npr = P_munu_packed.shape
    N = TA.shape[0]
    il = np.tril_indices(N)
    full = np.zeros((naux, N, N))
    full[:, il[0], il[1]] = P_munu_packed
    full[:, il[1], il[0]] = P_munu_packed
    Pij = np.einsum("Pmn,mi,nj->Pij", full, TA, TA, optimize=True)
    n = TA.shape[1]
    jl = np.tril_indices(n)
    sym = Pij[:, jl[0], jl[1]]
    X = scipy.linalg.solve_triangular(L_PQ, sym, lower=True)
    return X.T @ X


def transform_integral_semisparse(P_munu_packed, stored_pairs, TA, S_abs, L_PQ, MO_coeff_epsilon):
    """Literal restatement of transform_integral (_cpp/eri_sparse_DF.cpp:739-751):
    get_AO_per_MO (:443-465) -> contract_with_TA_1st (:484-532) -> contract_with_TA_2nd_to_sym_dense (:560-605) ->
    eval_via_cholesky (:611-621).  P_munu_packed: (naux, npair(N)) for mu >= nu; stored_pairs: boolean (N, N) mask of the
    AO pairs present in the SemiSparseSym3DTensor (exch_reachable)."""
    import scipy.linalg
    naux = P_munu_packed.shape[0]
    N, nmo = TA.shape
    X = np.abs(S_abs @ TA)
    AO_by_MO = [[mu for mu in range(N) if X[mu, i] >= MO_coeff_epsilon] for i in range(nmo)]
    col = lambda mu, nu: ravel_symmetric(mu, nu)
    g = {}
    for i in range(nmo):
        for mu in AO_by_MO[i]:
            acc = np.zeros(naux)
            for nu in range(N):
                if stored_pairs[mu, nu]:
                    acc += TA[nu, i] * P_munu_packed[:, col(mu, nu)]
            g[(mu, i)] = acc
    npr = npair(nmo)
    sym = np.zeros((naux, npr))
    for ij in range(npr):
        # unravel_symmetric (indexers.hpp:89-96) returns (smaller, larger): the AO list of the SMALLER orbital index is walked
        j = int((np.sqrt(8 * ij + 1) - 1) // 2)
        i = ij - j * (j + 1) // 2
        assert i <= j
        tmp = np.zeros(naux)
        for mu in AO_by_MO[i]:
            tmp += TA[mu, j] * g[(mu, i)]
        sym[:, ij] = tmp
    Xs = scipy.linalg.solve_triangular(L_PQ, sym, lower=True)
    return Xs.T @ Xs


def transform_integral_semisparse_csr(unique_dense_data, exch_reachable_with_offsets, TA, S_abs, L_PQ, MO_coeff_epsilon):
    """transform_integral (_cpp/eri_sparse_DF.cpp:739-751) on the reference's own storage: `unique_dense_data` (naux, n_unique),
    one aux vector per stored unique AO pair, and `exch_reachable_with_offsets[mu] = [(column, nu), ...]` (:260-279).
    The loops are those of get_AO_per_MO (:443-465), contract_with_TA_1st (:514-521), contract_with_TA_2nd_to_sym_dense (:586-595,
    (i, j) = unravel_symmetric -> i <= j, the AO list of i is walked) and eval_via_cholesky (:611-621)."""
    import scipy.linalg
    naux = unique_dense_data.shape[0]
    N, nmo = TA.shape
    X = np.abs(S_abs @ TA) if S_abs is not None else None
    AO_by_MO = [[mu for mu in range(N) if X is None or X[mu, i] >= MO_coeff_epsilon] for i in range(nmo)]
    g = {}
    for i in range(nmo):
        for mu in AO_by_MO[i]:
            acc = np.zeros(naux)
            for off, nu in exch_reachable_with_offsets[mu]:
                acc += TA[nu, i] * unique_dense_data[:, off]
            g[(mu, i)] = acc
    npr = npair(nmo)
    sym = np.zeros((naux, npr))
    for ij in range(npr):
        j = int((np.sqrt(8 * ij + 1) - 1) // 2)
        i = ij - j * (j + 1) // 2
        tmp = np.zeros(naux)
        for mu in AO_by_MO[i]:
            tmp += TA[mu, j] * g[(mu, i)]
        sym[:, ij] = tmp
    Xs = scipy.linalg.solve_triangular(L_PQ, sym, lower=True)
    return Xs.T @ Xs
