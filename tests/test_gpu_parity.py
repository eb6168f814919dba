"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle, the reference's golden vectors
and size-independent properties.  Tolerances: integrals 1e-12 absolute (north star asks 1e-8 Eh on energies),
J/K 1e-10 absolute on O(10) values, energies 1e-9 Eh."""
import numpy as np
import pytest

from conftest import R_N2, atom_arrays, make_system
from oracle import oracle as orc
from oracle import scf_oracle as so
from tuna_amd import molecule as mol

pytestmark = pytest.mark.gpu

TOL_INT = 1e-12


@pytest.mark.parametrize("tag", ["h2_sto3g_1p4", "h2_sto3g", "n2_sto3g", "he_631g"])
def test_small_systems_against_reference_golden(engine, small, tag):
    atoms, shells, aos, _ = make_system(tag)
    g = small[tag]
    engine.set_basis(aos)
    norm, coefs = engine.norms()
    np.testing.assert_allclose(norm, g["norm"], rtol=1e-15)
    np.testing.assert_allclose(coefs, g["coefs"], rtol=1e-15)
    np.testing.assert_allclose(engine.sph_matrix(), g["U"], atol=3e-16)
    engine.build_eri(spherical=False)
    E = engine.copy_eri()
    assert np.abs(E - g["ERI"]).max() < TOL_INT
    xyz, chg, org = atom_arrays(atoms)
    for got, name in zip(engine.one_electron(xyz, chg, org, spherical=False), "STVDQ"):
        assert np.abs(got - g[name]).max() < TOL_INT, name


@pytest.mark.parametrize("tag", ["n2_ccpvdz", "c2_n2_ccpvtz", "c4_co_def2tzvp", "high_l"])
def test_full_tensor_against_oracle(engine, golden, tag):
    """Every element of the Cartesian and the spherical tensor vs the C oracle; samples vs the reference golden."""
    atoms, shells, aos, _ = make_system(tag)
    g = golden(tag)
    engine.set_basis(aos)
    Eo = orc.eri(aos)
    engine.build_eri(spherical=False)
    Ec = engine.copy_eri()
    assert np.abs(Ec - Eo).max() < TOL_INT
    # exact zeros where the x- or y-parity is odd (pyx:1324-1327)
    lx = aos.lmn[:, 0]; ly = aos.lmn[:, 1]
    odd = ((lx[:, None, None, None] + lx[None, :, None, None] + lx[None, None, :, None] + lx[None, None, None, :]) % 2 == 1) | \
          ((ly[:, None, None, None] + ly[None, :, None, None] + ly[None, None, :, None] + ly[None, None, None, :]) % 2 == 1)
    assert np.all(Ec[odd] == 0.0)
    idx = g["eri_idx"]
    assert np.abs(engine.sample_eri(idx) - g["eri_val"]).max() < TOL_INT
    del Ec
    U = engine.sph_matrix()
    np.testing.assert_allclose(U, g["U"], atol=3e-16)
    engine.build_eri(spherical=True)
    Es = engine.copy_eri()
    Eos = so.eri_to_spherical(U, Eo)
    assert np.abs(Es - Eos).max() < TOL_INT
    assert np.abs(engine.sample_eri(g["eri_sph_idx"]) - g["eri_sph_val"]).max() < TOL_INT
    assert abs(np.sqrt(np.sum(Es * Es)) - g["eri_sph_fro"]) < 1e-10 * g["eri_sph_fro"]
    # the dense copy carries all eight images
    for perm in [(1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1), (3, 2, 1, 0)]:
        assert np.abs(Es - Es.transpose(perm)).max() < 1e-13
    # one-electron matrices, spherical
    xyz, chg, org = atom_arrays(atoms)
    for got, name in zip(engine.one_electron(xyz, chg, org, spherical=True), "STV"):
        assert np.abs(got - so.to_spherical(g["U"], g[name])).max() < TOL_INT, name
    # J/K vs the reference einsums
    if "P_rand" in g.files:
        J, K = engine.fock_jk(g["P_rand"])
        assert np.abs(J - g["J_rand"]).max() < 1e-10
        assert np.abs(K - g["K_rand"]).max() < 1e-10
    rng = np.random.default_rng(3)
    A = rng.standard_normal((engine.N, engine.N))
    P = A + A.T
    J, K = engine.fock_jk(P)
    assert np.abs(J - so.coulomb(P, Eos)).max() < 1e-10 and np.abs(K - so.exchange(P, Eos)).max() < 1e-10


def test_c3_ar2_ccpvqz_samples_and_jk(engine, golden):
    """C3 (148 Cartesian / 118 spherical AOs, 13-primitive s contractions, g shells): samples of the reference tensor."""
    atoms, shells, aos, _ = make_system("c3_ar2_ccpvqz")
    g = golden("c3_ar2_ccpvqz")
    engine.set_basis(aos)
    engine.build_eri(spherical=False)
    assert np.abs(engine.sample_eri(g["eri_idx"]) - g["eri_val"]).max() < TOL_INT
    engine.build_eri(spherical=True)
    assert np.abs(engine.sample_eri(g["eri_sph_idx"]) - g["eri_sph_val"]).max() < TOL_INT
    J, K = engine.fock_jk(g["P_rand"])
    assert np.abs(J - g["J_rand"]).max() < 1e-9
    assert np.abs(K - g["K_rand"]).max() < 1e-9
    xyz, chg, org = atom_arrays(atoms)
    for got, name in zip(engine.one_electron(xyz, chg, org, spherical=False), "STVDQ"):
        assert np.abs(got - g[name]).max() < 1e-11, name


def test_odd_dimension_padding_and_multi_density(engine):
    """N odd -> padded leading dimension; several densities in one call."""
    basis = {1: [("S", [(1.1, 1.0)]), ("P", [(0.8, 1.0)])], 2: [("S", [(2.0, 0.5), (0.6, 0.6)]), ("D", [(1.0, 1.0)]), ("S", [(0.3, 1.0)])]}
    atoms = mol.make_atoms(["H", "HE"], 1.7)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, basis))
    engine.set_basis(aos).build_eri(True)
    assert engine.N % 2 == 1 and engine.eri_storage()["ld"] == engine.N + 1
    Es = so.eri_to_spherical(engine.sph_matrix(), orc.eri(aos))
    assert np.abs(engine.copy_eri() - Es).max() < TOL_INT
    rng = np.random.default_rng(5)
    P = rng.standard_normal((3, engine.N, engine.N))
    J, K = engine.fock_jk(P)
    for d in range(3):   # non-symmetric densities too: the einsum definitions are what is implemented
        assert np.abs(J[d] - so.coulomb(P[d], Es)).max() < 1e-11
        assert np.abs(K[d] - so.exchange(P[d], Es)).max() < 1e-11


def test_decontracted_ao_order_falls_back_to_single_component_shells(engine):
    """The reference's DECONTRACT emits, per Cartesian component, one AO per primitive (tuna_molecule.py:564-570);
    such a list has no contiguous shells, is handled AO by AO, and is Cartesian-only."""
    atoms = mol.make_atoms(["N", "N"], R_N2)
    tbl = mol.atomic_basis("6-31G*", 7)
    origin, lmn, nprim, exps, coefs = [], [], [], [], []
    for at in atoms:
        for letter, prims in tbl:
            L = mol.SHELL_LETTERS.find(letter)
            for comp in mol.cartesian_components(L):
                for e, _ in prims:
                    origin.append(at.origin); lmn.append(comp); nprim.append(1); exps.append([e]); coefs.append([1.0])
    off = np.zeros(len(nprim) + 1, dtype=np.int32); off[1:] = np.cumsum(nprim)
    aos = mol.AOList(np.array(origin, dtype=float), np.array(lmn, dtype=np.int32), np.array(nprim, dtype=np.int32), off,
                     np.concatenate(exps), np.concatenate(coefs))
    engine.set_basis(aos).build_eri(spherical=False)
    assert np.abs(engine.copy_eri() - orc.eri(aos)).max() < TOL_INT
    from tuna_amd._lib import TunaError
    with pytest.raises(TunaError):
        engine.build_eri(spherical=True)


def test_single_integral_and_cross_overlap(engine):
    atoms, shells, aos, _ = make_system("n2_ccpvdz")
    rng = np.random.default_rng(11)
    Eo = None
    for _ in range(4):
        q = rng.integers(0, aos.n, 4)
        sub = mol.AOList(aos.origin[q], aos.lmn[q], aos.nprim[q], np.concatenate([[0], np.cumsum(aos.nprim[q])]).astype(np.int32),
                         np.concatenate([aos.exps[aos.prim_off[i]:aos.prim_off[i + 1]] for i in q]),
                         np.concatenate([aos.coefs[aos.prim_off[i]:aos.prim_off[i + 1]] for i in q]))
        if Eo is None:
            Eo = orc.eri(aos)
        assert abs(engine.eri_element(sub) - Eo[q[0], q[1], q[2], q[3]]) < TOL_INT
    _, _, aos2, _ = make_system("n2_sto3g")
    engine.set_basis(aos)
    assert np.abs(engine.cross_overlap(aos2) - orc.cross_overlap(aos, aos2)).max() < TOL_INT


def test_error_behaviour(engine):
    from tuna_amd._lib import TunaError
    atoms, shells, aos, _ = make_system("h2_sto3g")
    bad = mol.AOList(aos.origin.copy(), aos.lmn, aos.nprim, aos.prim_off, aos.exps, aos.coefs)
    bad.origin[1, 0] = 0.3          # off the z axis (kernel:386-388)
    with pytest.raises(TunaError) as e:
        engine.set_basis(bad)
    assert "aligned" in str(e.value) and e.value.code == -6
    engine.set_basis(aos)
    with pytest.raises(TunaError):
        engine.fock_jk(np.eye(2))   # no tensor built yet


@pytest.mark.parametrize("n_sph", [120, 200, 400])
def test_synthetic_series_properties(engine, n_sph):
    """Synthetic even-tempered Ar2-like diatomic: properties that hold at any size -- up to the 400-AO
    workload of bench.py (27 GB of stored tensor)."""
    counts = mol.synthetic_counts(n_sph)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
    engine.set_basis(aos).build_eri(True)
    N = engine.N
    assert N == n_sph
    rng = np.random.default_rng(0)
    idx = rng.integers(0, N, size=(4000, 4)).astype(np.int32)
    v = engine.sample_eri(idx)
    for perm in [(1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1), (3, 2, 0, 1)]:     # 8-fold symmetry
        assert np.abs(v - engine.sample_eri(idx[:, perm])).max() < 1e-12
    diag = engine.sample_eri(np.stack([idx[:, 0], idx[:, 1], idx[:, 0], idx[:, 1]], axis=1))
    assert diag.min() > -1e-13                                              # (ij|ij) >= 0
    assert np.all(np.abs(v) <= np.sqrt(np.abs(diag) * np.abs(engine.sample_eri(np.stack([idx[:, 2], idx[:, 3], idx[:, 2], idx[:, 3]], axis=1)))) + 1e-10)  # Schwarz
    A = rng.standard_normal((N, N)); B = rng.standard_normal((N, N))
    P1, P2 = A + A.T, B + B.T
    J1, K1 = engine.fock_jk(P1)
    J2, K2 = engine.fock_jk(P2)
    J3, K3 = engine.fock_jk(2.0 * P1 - 0.5 * P2)
    scale = np.abs(J1).max()
    assert np.abs(J3 - (2.0 * J1 - 0.5 * J2)).max() < 1e-11 * scale          # linearity
    assert np.abs(K3 - (2.0 * K1 - 0.5 * K2)).max() < 1e-11 * scale
    assert np.abs(J1 - J1.T).max() < 1e-11 * scale and np.abs(K1 - K1.T).max() < 1e-11 * scale
    assert abs(np.sum(P2 * J1) - np.sum(P1 * J2)) < 1e-9 * abs(np.sum(P2 * J1))   # <P2|J[P1]> = <P1|J[P2]>
    assert abs(np.sum(P2 * K1) - np.sum(P1 * K2)) < 1e-9 * abs(np.sum(P2 * K1))
    # a sample of rows of J and K against a direct contraction of sampled tensor rows
    i, j = 7, 3
    rows = np.array([[i, j, k, l] for k in range(N) for l in range(N)], dtype=np.int32)
    M = engine.sample_eri(rows).reshape(N, N)
    assert abs(J1[i, j] - np.sum(M * P1)) < 1e-10 * scale


def test_fock_rows_at_the_benched_size_against_sampled_tensor_rows(engine):
    """N = 400, the tensor bench.py times: 24 elements each of J and K -- AOs of all four parity classes, the first and last AOs, both
    triangles -- against a direct contraction of sampled tensor rows with the reference's index strings (scf:70 "ijkl,kl->ij": J_ab =
    sum_kl (ab|kl) P_kl; scf:42 "ilkj,kl->ij": K_ab = sum_kl (al|kb) P_kl), for the one-density pass and for the fused two-density pass.
    (The tensor itself is pinned at this size by test_bench_workload_tensor_against_oracle.)"""
    n_sph = 400
    counts = mol.synthetic_counts(n_sph)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
    engine.set_basis(aos).build_eri(True)
    N = engine.N
    rng = np.random.default_rng(17)
    A = rng.standard_normal((2, N, N))
    P = A + A.transpose(0, 2, 1)
    J1, K1 = engine.fock_jk(P[0])
    J2, K2 = engine.fock_jk(P)                                            # both densities in one pass
    assert np.array_equal(J2[0], J1) or np.abs(J2[0] - J1).max() < 1e-11 * np.abs(J1).max()
    picks = [(0, 0), (N - 1, N - 1), (N - 1, 0), (0, N - 1), (N // 2, N // 2 - 1), (199, 200), (200, 199), (399, 200)]
    while len(picks) < 24:
        a, b = (int(x) for x in rng.integers(0, N, size=2))
        picks.append((a, b))
    kk, ll = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    kk, ll = kk.reshape(-1).astype(np.int32), ll.reshape(-1).astype(np.int32)
    sJ, sK = np.abs(J1).max(), np.abs(K1).max()
    for (a, b) in picks:
        ia, ib = np.full(N * N, a, dtype=np.int32), np.full(N * N, b, dtype=np.int32)
        Mj = engine.sample_eri(np.stack([ia, ib, kk, ll], axis=1)).reshape(N, N)       # (ab|kl)
        Mk = engine.sample_eri(np.stack([ia, ll, kk, ib], axis=1)).reshape(N, N)       # (al|kb) at [k][l]
        for d, (J, K) in enumerate(((J1, K1), (J2[1], K2[1]))):
            assert abs(J[a, b] - np.sum(Mj * P[d])) < 1e-10 * sJ, (a, b, d)
            assert abs(K[a, b] - np.sum(Mk * P[d])) < 1e-10 * sK, (a, b, d)


def test_multi_chunk_contraction_against_the_reference_einsums(engine):
    """The contraction itself where a parity class is wider than one 64-column chunk (N = 200: 81 / 48 / 48 / 23 AOs per class, so the
    tasks of jk_packed_kernel come in several chunks per class -- no golden system reaches that): the dense copy of the GPU tensor (pinned
    block-wise to the oracle by test_bench_workload_tensor_against_oracle / test_synthetic_series_properties) goes through the
    reference's own einsum strings (scf:70, scf:42) in NumPy; J and K of the one-density pass, of the fused two-density pass and of a
    non-symmetric density (two passes) must agree to 1e-10 relative."""
    n_sph = 200
    counts = mol.synthetic_counts(n_sph)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
    engine.set_basis(aos).build_eri(True, layout="packed")
    N = engine.N
    assert N == n_sph
    ERI = engine.copy_eri()                                               # dense [N,N,N,N], all images and the parity zeros
    rng = np.random.default_rng(11)
    A = rng.standard_normal((3, N, N))
    P = np.stack([A[0] + A[0].T, A[1] + A[1].T])
    M2 = ERI.reshape(N * N, N * N)

    def ref_jk(D):
        J = (M2 @ D.reshape(-1)).reshape(N, N)                            # np.einsum("ijkl,kl->ij", ERI, D)
        K = np.einsum("ilkj,kl->ij", ERI, D, optimize=True)
        return J, K
    refs = [ref_jk(P[0]), ref_jk(P[1]), ref_jk(A[2])]
    assert np.abs(refs[0][0] - so.coulomb(P[0], ERI)).max() < 1e-9 * np.abs(refs[0][0]).max()    # (the GEMV is the reference string "ijkl,kl->ij")
    J1, K1 = engine.fock_jk(P[0])                                         # jk_packed_kernel<1>
    J2, K2 = engine.fock_jk(P)                                            # jk_packed_kernel<2>: both densities in one pass
    J3, K3 = engine.fock_jk(A[2])                                         # non-symmetric: two passes, K = D(P^T) + D(P)^T
    for got, ref in ((J1, refs[0][0]), (K1, refs[0][1]), (J2[0], refs[0][0]), (K2[0], refs[0][1]), (J2[1], refs[1][0]), (K2[1], refs[1][1]),
                     (J3, refs[2][0]), (K3, refs[2][1])):
        assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()
    # the pass variants share partial-sum buffers whose never-written entries must stay zero (the reductions read them): after the
    # two-density and the two-pass builds, the one-density build still gives bitwise what it gave on the fresh buffers
    J1b, K1b = engine.fock_jk(P[0])
    assert np.array_equal(J1, J1b) and np.array_equal(K1, K1b)


def test_packed_and_rows_layouts_agree(engine):
    """The 8-fold packed tensor + jk_packed_kernel against the (i >= j) x [k][l] rows layout + jk_rows_kernel: same tensor values,
    same J and K for one and for two densities (the fused two-density pass included)."""
    counts = mol.synthetic_counts(200)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
    rng = np.random.default_rng(3)
    N = 200
    A = rng.standard_normal((2, N, N)); P = A + A.transpose(0, 2, 1)
    idx = rng.integers(0, N, size=(5000, 4)).astype(np.int32)
    res = {}
    for layout in ("rows", "packed"):
        engine.set_basis(aos).build_eri(True, layout=layout)
        st = engine.eri_storage()
        assert st["layout"] == layout
        J1, K1 = engine.fock_jk(P[0])
        J2, K2 = engine.fock_jk(P)                                   # two densities in one call
        res[layout] = (engine.sample_eri(idx), J1, K1, J2, K2, st["bytes"])
    a, b = res["rows"], res["packed"]
    assert np.abs(a[0] - b[0]).max() < 1e-13
    scale = np.abs(a[1]).max()
    for q in (1, 2, 3, 4):
        assert np.abs(a[q] - b[q]).max() < 1e-11 * scale
    assert np.abs(b[3][0] - b[1]).max() < 1e-11 * scale and np.abs(b[4][0] - b[2]).max() < 1e-11 * scale   # fused pass = single pass
    assert b[5] < 0.30 * a[5]                                         # ~N^4 bytes (+ cache-line padding: 11 % at N = 200) against 4 N^4


def test_rebuilding_reuses_the_tensor_buffer(engine, small):
    """The tensor buffer is kept across builds (tf_build_eri): a larger basis, then a smaller one in the same buffer, then the
    larger one again -- every build is complete (no stale values), rebuilds are bitwise identical."""
    aos_big, aos_small = make_system("n2_ccpvdz")[2], make_system("n2_sto3g")[2]
    rng = np.random.default_rng(5)
    nb = engine.set_basis(aos_big).build_eri(True).N
    idx = rng.integers(0, nb, size=(4000, 4)).astype(np.int32)
    first = engine.sample_eri(idx)
    bytes_big = engine.eri_storage()["bytes"]
    engine.set_basis(aos_small).build_eri(True)
    assert engine.eri_storage()["bytes"] < bytes_big
    engine.build_eri(spherical=False)                                 # (the golden tensor of this system is Cartesian)
    assert np.abs(engine.copy_eri() - small["n2_sto3g"]["ERI"]).max() < TOL_INT
    engine.set_basis(aos_big).build_eri(True)
    assert np.array_equal(engine.sample_eri(idx), first)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_contracted_bases_against_oracle(engine, seed):
    """Seeded random basis sets -- shells s..g with 1-5 primitives each, different on the two atoms, also a single atom -- through
    the small-problem path (contracted and uncontracted quartets of every angular-momentum group in one build): every element of
    the spherical tensor against the C oracle, and a Fock build against the reference's einsums."""
    rng = np.random.default_rng(seed)

    def random_atom_basis(n_shells, l_max):
        out = []
        for _ in range(n_shells):
            L = int(rng.integers(0, l_max + 1))
            nprim = int(rng.integers(1, 6)) if L <= 2 else int(rng.integers(1, 3))
            exps = np.sort(10.0 ** rng.uniform(-1.0, 1.6, nprim))[::-1]
            coefs = rng.uniform(0.2, 1.0, nprim) * rng.choice([-1.0, 1.0], nprim)
            out.append(("SPDFG"[L], [(float(e), float(c)) for e, c in zip(exps, coefs)]))
        return out

    symbols, R = ((["N", "O"], 2.0 + 0.5 * rng.random()) if seed != 13 else (["N"], None))
    basis = {7: random_atom_basis(5, 4), 8: random_atom_basis(4, 3)}
    atoms = mol.make_atoms(symbols, R)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, basis))
    engine.set_basis(aos)
    Eo = orc.eri(aos)
    scale = np.abs(Eo).max()
    engine.build_eri(spherical=False)
    assert np.abs(engine.copy_eri() - Eo).max() < 1e-12 * max(1.0, scale)
    U = engine.sph_matrix()
    engine.build_eri(spherical=True)
    Es = engine.copy_eri()
    Eos = so.eri_to_spherical(U, Eo)
    assert np.abs(Es - Eos).max() < 1e-12 * max(1.0, scale)
    A = rng.standard_normal((engine.N, engine.N))
    P = A + A.T
    J, K = engine.fock_jk(P)
    tol = 1e-10 * max(1.0, scale) * engine.N
    assert np.abs(J - so.coulomb(P, Eos)).max() < tol and np.abs(K - so.exchange(P, Eos)).max() < tol


def _shell_quartet_blocks_against_oracle(engine, shells, n_quartets, seed, spherical):
    """Random shell quartets (A B|C D) of the resident tensor, every element of each block, against the C oracle run on the four
    shells alone (the oracle restates the reference per AO quartet, so a four-shell basis gives exactly the tensor's values)."""
    from tuna_amd import spherical as sph
    rng = np.random.default_rng(seed)
    dim = [(s.n_sph if spherical else s.n_cart) for s in shells]
    off = np.concatenate([[0], np.cumsum(dim)])
    worst, n_checked = 0.0, 0
    for _ in range(n_quartets):
        q = [int(x) for x in rng.integers(0, len(shells), size=4)]
        sub = [shells[x] for x in q]
        aos4 = mol.expand_cartesian_aos(sub)
        Eo = orc.eri(aos4, 2)
        nc = [s.n_cart for s in sub]
        co = np.concatenate([[0], np.cumsum(nc)])
        blk = Eo[co[0]:co[1], co[1]:co[2], co[2]:co[3], co[3]:co[4]]
        if spherical:
            U = [sph.spherical_block(s.L) for s in sub]
            blk = np.einsum("ia,jb,kc,ld,abcd->ijkl", U[0], U[1], U[2], U[3], blk, optimize=True)
        idx = np.stack(np.meshgrid(*[np.arange(off[x], off[x + 1]) for x in q], indexing="ij"), axis=-1).reshape(-1, 4).astype(np.int32)
        got = engine.sample_eri(idx).reshape(blk.shape)
        worst = max(worst, float(np.abs(got - blk).max()))
        n_checked += blk.size
    return worst, n_checked


def test_bench_workload_tensor_against_oracle(engine):
    """The 400-AO synthetic tensor of bench.py itself (per-class ERI kernels: eri_fact_kernel / eri_multi_kernel, fused ket transform,
    parity-blocked rows) pinned to the oracle: every element of 150 random shell-quartet blocks (s..f shells, both atoms)."""
    counts = mol.synthetic_counts(400)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    shells = mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)})
    engine.set_basis(mol.expand_cartesian_aos(shells)).build_eri(True)
    assert engine.N == 400
    worst, n = _shell_quartet_blocks_against_oracle(engine, shells, 150, 21, True)
    assert n >= 2000 and worst < TOL_INT, (worst, n)


def test_per_class_kernels_cartesian_tensor_against_oracle(engine, monkeypatch):
    """The per-class kernels on ONE context with Cartesian output (no spherical transform in between): synthetic 200-AO basis."""
    monkeypatch.setenv("TF_ERI_MODE", "class")
    counts = mol.synthetic_counts(200)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    shells = mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)})
    engine.set_basis(mol.expand_cartesian_aos(shells)).build_eri(False)
    worst, n = _shell_quartet_blocks_against_oracle(engine, shells, 120, 22, False)
    assert n >= 2000 and worst < TOL_INT, (worst, n)


@pytest.mark.parametrize("tag", ["high_l", "c4_co_def2tzvp"])
@pytest.mark.parametrize("mode", ["class", "generic"])
def test_full_tensor_both_eri_modes(engine, monkeypatch, tag, mode):
    """Every element of the spherical and Cartesian tensors with the ERI launch mode forced (TF_ERI_MODE): the per-class kernels
    (eri_class_kernel for the contracted classes, eri_fact / eri_multi for the uncontracted ones) and the small-problem kernel
    (eri_cfact_kernel) must agree with the oracle on the same system, one context."""
    monkeypatch.setenv("TF_ERI_MODE", mode)
    atoms, shells, aos, _ = make_system(tag)
    engine.set_basis(aos)
    Eo = orc.eri(aos)
    engine.build_eri(spherical=False)
    assert np.abs(engine.copy_eri() - Eo).max() < TOL_INT
    U = engine.sph_matrix()
    engine.build_eri(spherical=True)
    Eos = so.eri_to_spherical(U, Eo)
    assert np.abs(engine.copy_eri() - Eos).max() < TOL_INT
    rng = np.random.default_rng(8)
    A = rng.standard_normal((engine.N, engine.N))
    P = A + A.T
    J, K = engine.fock_jk(P)
    assert np.abs(J - so.coulomb(P, Eos)).max() < 1e-10 and np.abs(K - so.exchange(P, Eos)).max() < 1e-10


@pytest.mark.parametrize("env", [{"TF_ERI_FAMILIES": "1"}, {"TF_ERI_FAMILIES": "1", "TF_ERI_CC_FAMILIES": "0"},
                                 {"TF_ERI_FAMILIES": "1", "TF_ERI_BRA_FAMILIES": "1"}, {"TF_ERI_FAMILIES": "0"}])
@pytest.mark.parametrize("tag", ["c2_n2_ccpvtz", "c4_co_def2tzvp"])
def test_families_of_generally_contracted_shell_pairs(engine, monkeypatch, tag, env):
    """eri_cfact_kernel with families of shell pairs over shells that repeat one primitive list (general contractions: N cc-pVTZ keeps two s
    shells on the same 8 primitives; def2-TZVP has none and must be left alone): ket families (1 x 9), families on both sides of the
    contracted-against-contracted launches (3 x 9), bra families (9 x 1, off by default), and everything off -- every element of the
    spherical tensor against the oracle.  (By default families start at 8e6 primitive shell quartets: BASELINE config 3, Ar2/cc-pVQZ, runs
    them in test_golden_samples / the sharded tests; here they are forced on small systems whose whole tensor the oracle can afford.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    atoms, shells, aos, _ = make_system(tag)
    engine.set_basis(aos)
    Eos = so.eri_to_spherical(engine.sph_matrix(), orc.eri(aos))
    engine.build_eri(spherical=True)
    assert np.abs(engine.copy_eri() - Eos).max() < TOL_INT


def test_alternating_one_and_two_density_passes(engine, golden):
    """The partial-sum buffers are zeroed once per build and the reductions add every slot without a validity test: one- and
    two-density passes (different row groups, separate buffer regions) must not disturb each other, in any order, on one build."""
    for tag in ("n2_ccpvdz", "c2_n2_ccpvtz"):
        g = golden(tag)
        atoms, shells, aos, nocc = make_system(tag)
        engine.set_basis(aos).build_eri(True)
        A = np.random.default_rng(4).standard_normal((2, engine.N, engine.N))
        P = A + A.transpose(0, 2, 1)
        J1, K1 = engine.fock_jk(P[0])
        J2, K2 = engine.fock_jk(P)                    # two densities in one pass (groups of 4 rows)
        J3, K3 = engine.fock_jk(P[0])
        J4, K4 = engine.fock_jk(P[::-1].copy())
        for X, Y in ((J2[0], J1), (K2[0], K1), (J3, J1), (K3, K1), (J4[1], J1), (K4[1], K1), (J4[0], J2[1]), (K4[0], K2[1])):
            assert np.abs(X - Y).max() < 1e-10
        J, K = engine.fock_jk(g["P_rand"])
        assert np.abs(J - g["J_rand"]).max() < 1e-10 and np.abs(K - g["K_rand"]).max() < 1e-10


@pytest.mark.parametrize("parts", [2, 3, 5])
def test_cut_walks_give_the_same_fock_matrices(engine, golden, monkeypatch, parts):
    """Several ranks cut the k walks of the Fock kernel into parts (shorter tasks: TF_JK_PARTS forces it on one rank): one plane of
    column parts and of Jd per part, the segment of k == i in the last part -- same J and K, for one and for two densities."""
    monkeypatch.setenv("TF_JK_PARTS", str(parts))
    for tag in ("n2_ccpvdz", "c2_n2_ccpvtz"):
        g = golden(tag)
        atoms, shells, aos, nocc = make_system(tag)
        engine.set_basis(aos).build_eri(True)
        J, K = engine.fock_jk(g["P_rand"])
        assert np.abs(J - g["J_rand"]).max() < 1e-10 and np.abs(K - g["K_rand"]).max() < 1e-10
        P2 = np.stack([g["P_rand"], 0.5 * g["P_rand"] + 0.25 * np.diag(np.diag(g["P_rand"]))])
        J2, K2 = engine.fock_jk(P2)
        assert np.abs(J2[0] - J).max() < 1e-10 and np.abs(K2[0] - K).max() < 1e-10
        J1, K1 = engine.fock_jk(P2[1])
        assert np.abs(J2[1] - J1).max() < 1e-10 and np.abs(K2[1] - K1).max() < 1e-10
