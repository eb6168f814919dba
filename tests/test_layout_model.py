"""CPU: the NumPy model of the parity-blocked packed layout (tests/layout_model.py) -- the tables tf_build_eri builds, the task
structure of jk_packed_kernel and the validity rules of its reductions -- against the reference einsums (scf:55-72, scf:27-44) on
random tensors with the 8-fold symmetry and the x/y parity zeros of a z-axis diatomic (pyx:1324-1327)."""
import numpy as np
import pytest

import layout_model as lm

SPH_CLASSES = {0: [0], 1: [1, 2, 0], 2: [3, 1, 2, 0, 0], 3: [2, 3, 2, 0, 1, 0, 1]}     # parity class of every real harmonic, reference order


def classes_of(shell_L):
    out = []
    for L in shell_L:
        out += SPH_CLASSES[L]
    return out


@pytest.mark.parametrize("shell_L,pad,cw,jbb,w", [([0, 0, 1, 2, 1, 0, 2, 3, 1, 0], 4, 8, 4, 2), ([0, 1, 1, 2, 0, 1, 2, 2, 3, 0, 1], 8, 16, 8, 4),
                                                   ([0, 0, 0, 0], 8, 128, 8, 4), ([1, 0, 2], 2, 4, 2, 1)])
def test_model_reproduces_the_reference_einsums(shell_L, pad, cw, jbb, w):
    cls = classes_of(shell_L)
    N = len(cls)
    lm.JBB, lm.W = jbb, w
    try:
        L = lm.Layout(cls, pad, cw)
        E = lm.random_parity_tensor(cls, 1)
        A = np.random.default_rng(2).standard_normal((N, N))
        P = A + A.T
        Jref = np.einsum("ijkl,kl->ij", E, P)
        Kref = np.einsum("ilkj,kl->ij", E, P)
        rows = [(i, j) for i in range(N) for j in range(i + 1)]
        J, K, info = lm.fock_partial(L, E, P, rows)
        assert np.abs(J - Jref).max() < 1e-12 and np.abs(K - Kref).max() < 1e-12
        # two ranks with an arbitrary split of the rows: partial sums add up
        Js = Ks = 0.0
        for r in range(2):
            Jr, Kr, _ = lm.fock_partial(L, E, P, [(i, j) for (i, j) in rows if (7 * i + 3 * j) % 2 == r])
            Js, Ks = Js + Jr, Ks + Kr
        assert np.abs(Js - Jref).max() < 1e-12 and np.abs(Ks - Kref).max() < 1e-12
    finally:
        lm.JBB, lm.W = 8, 8


def test_stored_fraction():
    """20s15p13d10f per atom (the 400-AO bench workload): the blocked rows hold ~1/4 of the 8-fold unique tensor."""
    cls = classes_of(([0] * 20 + [1] * 15 + [2] * 13 + [3] * 10) * 2)
    L = lm.Layout(cls, 8, 64)
    assert L.N == 400 and L.NW == 8 and sorted(L.csize) == [46, 96, 96, 162]
    stored = sum(L.row_len(i, 0) * (i + 1) for i in range(0))         # (row_len depends on the class of j: sum per class below)
    tot = 0
    for i in range(L.N):
        per_class = [L.secoff(c, i)[1] for c in range(4)]
        for j in range(i + 1):
            tot += per_class[L.cls[i] ^ L.cls[j]]
    assert 6.9e9 < 8 * tot < 7.2e9                                       # bytes; the unblocked packed layout stored 27.0e9
