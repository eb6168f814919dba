"""CPU: the NumPy model of the "tiles" tensor layout (tests/tile_model.py) on the tables of the library's own host builder
(tuna_amd/csrc/tf_tiles_host.h) against the reference einsums (scf:55-72, scf:27-44) on random tensors with the 8-fold symmetry and
the x/y parity zeros of a z-axis diatomic (pyx:1324-1327)."""
import numpy as np
import pytest

import tile_model as tm

SPH_CLASSES = {0: [0], 1: [1, 2, 0], 2: [3, 1, 2, 0, 0], 3: [2, 3, 2, 0, 1, 0, 1]}     # parity class of every real harmonic, reference order


def classes_of(shell_L):
    out = []
    for L in shell_L:
        out += SPH_CLASSES[L]
    return out


def reference(cls, seed=1):
    N = len(cls)
    E = tm.random_parity_tensor(cls, seed)
    A = np.random.default_rng(seed + 1).standard_normal((N, N))
    P = A + A.T
    return E, P, np.einsum("ijkl,kl->ij", E, P), np.einsum("ilkj,kl->ij", E, P)


@pytest.mark.parametrize("shell_L,ksub,part_steps", [([0, 0, 1, 2, 1, 0, 2, 3, 1, 0], 64, 48), ([0, 1, 1, 2, 0, 1, 2, 2, 3, 0, 1], 64, 3),
                                                      ([0, 0, 0, 0], 64, 48), ([1, 0, 2], 16, 2), ([0, 1, 1, 2, 0, 1, 2, 2, 3, 0, 1], 16, 5),
                                                      ([0, 0, 1, 2, 1, 0, 2, 3, 1, 0], 32, 48)])
def test_model_reproduces_the_reference_einsums(shell_L, ksub, part_steps):
    cls = classes_of(shell_L)
    N = len(cls)
    E, P, Jref, Kref = reference(cls)
    rows = [(i, j) for i in range(N) for j in range(i + 1)]
    T = tm.Tables(cls, rows, ksub, part_steps)
    buf, written = tm.pack_tensor(T, E)
    assert written.sum() >= tm.stored_count(T)
    J, K = tm.fock(T, buf, P)
    assert np.abs(J - Jref).max() < 1e-11 and np.abs(K - Kref).max() < 1e-11


def test_many_s_functions_cross_strip_and_block_boundaries():
    # 70 s functions + a few p: one class with two strips of 64 rows and several column blocks (triangle and rectangles)
    cls = [0] * 70 + classes_of([1] * 5)
    rng = np.random.default_rng(5)
    cls = list(rng.permutation(cls))
    N = len(cls)
    E, P, Jref, Kref = reference(cls, 3)
    rows = [(i, j) for i in range(N) for j in range(i + 1)]
    for ksub, steps in ((64, 40), (16, 1000)):
        T = tm.Tables(cls, rows, ksub, steps)
        buf, _ = tm.pack_tensor(T, E)
        J, K = tm.fock(T, buf, P)
        assert np.abs(J - Jref).max() < 1e-10 and np.abs(K - Kref).max() < 1e-10
        assert T.bucket[0] == 0 and T.bucket[4] == T.n_tasks


def test_two_ranks_partial_sums_add_up():
    # rows split by whole first indices and by runs of j (what tf_shard_plan_pairs hands out): the partial J and K add up
    cls = classes_of([0, 1, 1, 2, 0, 1, 2, 2, 3, 0, 1])
    N = len(cls)
    E, P, Jref, Kref = reference(cls, 7)
    rows = [(i, j) for i in range(N) for j in range(i + 1)]
    for split in (lambda i, j: i % 2, lambda i, j: int(j >= i // 2)):
        Js = Ks = 0.0
        for r in range(2):
            T = tm.Tables(cls, [(i, j) for (i, j) in rows if split(i, j) == r], 64, 48)
            buf, _ = tm.pack_tensor(T, E)
            Jr, Kr = tm.fock(T, buf, P)
            Js, Ks = Js + Jr, Ks + Kr
        assert np.abs(Js - Jref).max() < 1e-11 and np.abs(Ks - Kref).max() < 1e-11


def test_stored_bytes_are_close_to_the_unique_elements():
    cls = classes_of([0] * 20 + [1] * 15 + [2] * 13 + [3] * 10)        # the layout of the benched 400-function basis (one atom's worth, twice)
    cls = cls + cls
    N = len(cls)
    rows = [(i, j) for i in range(N) for j in range(i + 1)]
    T = tm.Tables(cls, rows, 64, 48)
    ratio = T.n_elems / tm.stored_count(T)
    assert ratio < 1.06, ratio


@pytest.mark.parametrize("shell_L,part_steps", [([0, 1, 1, 2, 0, 1, 2, 2, 3, 0, 1], 3), ([0] * 70 + [1] * 5, 7)])
def test_address_function_reaches_every_stored_element_once(shell_L, part_steps):
    """tt_elem_addr (what the writer and tf_copy_eri / tf_sample_eri use) against the regions: every canonical element of the owned rows
    has its own slot, and the slot holds that element in the packed buffer of the model"""
    cls = classes_of(shell_L)
    N = len(cls)
    rng = np.random.default_rng(3)
    cls = list(rng.permutation(cls))
    rows = [(i, j) for i in range(N) for j in range(i + 1) if (i + 2 * j) % 5 != 0 or i == j]
    rows = [(i, j) for (i, j) in rows]
    try:
        T = tm.Tables(cls, rows, 64, part_steps)
    except ValueError:
        rows = [(i, j) for i in range(N) for j in range(i + 1)]
        T = tm.Tables(cls, rows, 64, part_steps)
    E = tm.random_parity_tensor(cls, 4)
    buf, written = tm.pack_tensor(T, E)
    s, o = T.sigma, T.origI
    owned = set(rows)
    seen = set()
    n = 0
    for i in range(N):
        for j in range(i + 1):
            for k in range(i + 1):
                for l in range(k + 1):
                    if k == i and l > j:
                        continue
                    if (cls[i] ^ cls[j]) != (cls[k] ^ cls[l]):
                        continue
                    ad = T.elem_addr(s[i], s[j], s[k], s[l])
                    if (i, j) not in owned:
                        assert ad == -1
                        continue
                    assert 0 <= ad < T.n_elems and ad not in seen, (i, j, k, l, ad)
                    seen.add(ad)
                    assert buf[ad] == E[i, j, k, l]
                    n += 1
    assert n == tm.stored_count(T)
