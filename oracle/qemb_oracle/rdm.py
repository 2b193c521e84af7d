"""Oracle for row a9 (unrelaxed CCSD density matrices).  Test infrastructure."""
import numpy as np


def make_rdm1_ccsd_t1(t1):
    """shared/external/ccsd_rdm.py:10-20 (== cc.ccsd_rdm.make_rdm1 with l1 = l2 = 0, solver.py:922-924)."""
    nocc, nvir = t1.shape
    nmo = nocc + nvir
    dm = np.zeros((nmo, nmo))
    dm[:nocc, nocc:] = t1
    dm[nocc:, :nocc] = t1.T
    dm[np.diag_indices(nocc)] += 2.0
    return dm


def make_rdm2_urlx(t1, t2, with_dm1=True):
    """shared/external/ccsd_rdm.py:23-55."""
    nocc, nvir = t1.shape
    nmo = nocc + nvir
    goovv = (np.einsum("ia,jb->ijab", t1, t1) + t2) * 0.5
    dovov = goovv.transpose(0, 2, 1, 3) * 2 - goovv.transpose(1, 2, 0, 3)
    dm2 = np.zeros((nmo,) * 4)
    dm2[:nocc, nocc:, :nocc, nocc:] = dovov
    dm2[:nocc, nocc:, :nocc, nocc:] += dovov.transpose(2, 3, 0, 1)
    dm2[nocc:, :nocc, nocc:, :nocc] = dm2[:nocc, nocc:, :nocc, nocc:].transpose(1, 0, 3, 2)
    if with_dm1:
        dm1 = make_rdm1_ccsd_t1(t1)
        dm1[np.diag_indices(nocc)] -= 2
        for i in range(nocc):
            dm2[i, i, :, :] += dm1 * 2
            dm2[:, :, i, i] += dm1 * 2
            dm2[:, i, i, :] -= dm1
            dm2[i, :, :, i] -= dm1.T
        for i in range(nocc):
            for j in range(nocc):
                dm2[i, i, j, j] += 4
                dm2[i, j, j, i] -= 2
    return dm2


def add_dm1_terms(dm2, dm1, nocc):
    """The with_dm1=True completion of a normal-ordered 2-RDM (PySCF cc/ccsd_rdm.py _make_rdm2, same statements as
    shared/external/ccsd_rdm.py:40-53): products of the correlation 1-RDM with the HF determinant + the HF 2-RDM."""
    dm2 = dm2.copy()
    dm1 = dm1.copy()
    dm1[np.diag_indices(nocc)] -= 2
    for i in range(nocc):
        dm2[i, i, :, :] += dm1 * 2
        dm2[:, :, i, i] += dm1 * 2
        dm2[:, i, i, :] -= dm1
        dm2[i, :, :, i] -= dm1.T
    for i in range(nocc):
        for j in range(nocc):
            dm2[i, i, j, j] += 4
            dm2[i, j, j, i] -= 2
    return dm2
