"""GPU: the "tiles" tensor layout (tuna_amd/csrc/tf_tiles.h) and its matrix-core Fock kernels (tf_jktile.hip.h) against the reference
contractions (scf:55-72 "ijkl,kl->ij", scf:27-44 "ilkj,kl->ij"), the golden tensors of the reference engine and the packed layout.
The layout is opt-in (tf_set_eri_layout(TF_LAYOUT_TILES)); every test hands the shared context back with the default layout."""
import numpy as np
import pytest

from conftest import R_N2
from tuna_amd import molecule as mol

pytestmark = pytest.mark.gpu


def _synthetic(n_sph):
    counts = mol.synthetic_counts(n_sph)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    return mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))


def _reset(engine):
    engine._check(engine._L.tf_set_eri_layout(engine._ctx, -1))


@pytest.mark.parametrize("tag", ["n2_ccpvdz", "c2_n2_ccpvtz"])
def test_tiles_tensor_and_fock_matrices_against_reference_golden(engine, golden, tag):
    """full tensor (every element, parity zeros included) and J, K of the seeded density against the reference's own outputs"""
    z = golden(tag)
    basis = "cc-pVDZ" if tag == "n2_ccpvdz" else "cc-pVTZ"
    atoms = mol.make_atoms(["N", "N"], R_N2)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, basis))
    try:
        engine.set_basis(aos).build_eri(False, layout="tiles")               # Cartesian tensor (CARTHARM)
        assert engine.eri_storage()["layout"] == "tiles"
        assert np.abs(engine.sample_eri(z["eri_idx"].astype(np.int32)) - z["eri_val"]).max() < 1e-12
        engine.build_eri(True, layout="tiles")                               # real spherical harmonics: the reference's default
        assert np.abs(engine.sample_eri(z["eri_sph_idx"].astype(np.int32)) - z["eri_sph_val"]).max() < 1e-12
        Es = engine.copy_eri()
        assert abs(np.sqrt(np.sum(Es * Es)) - z["eri_sph_fro"]) < 1e-10 * z["eri_sph_fro"]
        P = z["P_rand"]
        assert P.shape == (engine.N, engine.N)
        J, K = engine.fock_jk(P)
        assert np.abs(J - z["J_rand"]).max() < 1e-10 and np.abs(K - z["K_rand"]).max() < 1e-10
    finally:
        _reset(engine)


def test_tiles_against_the_reference_einsums_and_the_packed_layout(engine):
    """N = 200 (several strips / column blocks per class, triangles with diagonal blocks): the dense copy of the tiles tensor equals the
    packed one bit for bit; J and K of one density, of a non-symmetric density (two passes) and of 2, 3, 5 and 8 densities in one call
    (the wide pass: densities as columns of the matrix-core products) against the reference einsum strings in NumPy."""
    aos = _synthetic(200)
    N = 200
    rng = np.random.default_rng(21)
    A = rng.standard_normal((9, N, N))
    P = A[:8] + A[:8].transpose(0, 2, 1)
    try:
        engine.set_basis(aos).build_eri(True, layout="packed")
        ERIp = engine.copy_eri()
        engine.build_eri(True, layout="tiles")
        st = engine.eri_storage()
        assert st["layout"] == "tiles"
        ERI = engine.copy_eri()
        assert np.array_equal(ERI, ERIp)
        M2 = ERI.reshape(N * N, N * N)

        def ref_jk(D):
            return (M2 @ D.reshape(-1)).reshape(N, N), np.einsum("ilkj,kl->ij", ERI, D, optimize=True)
        refs = [ref_jk(P[d]) for d in range(8)]
        J1, K1 = engine.fock_jk(P[0])
        assert np.abs(J1 - refs[0][0]).max() < 1e-10 * np.abs(refs[0][0]).max() and np.abs(K1 - refs[0][1]).max() < 1e-10 * np.abs(refs[0][1]).max()
        Jn, Kn = engine.fock_jk(A[8])                                       # non-symmetric: K = D(P^T) + D(P)^T
        rJ, rK = ref_jk(A[8])
        assert np.abs(Jn - rJ).max() < 1e-10 * np.abs(rJ).max() and np.abs(Kn - rK).max() < 1e-10 * np.abs(rK).max()
        for nd in (2, 3, 5, 8):
            Jw, Kw = engine.fock_jk(P[:nd])
            for d in range(nd):
                assert np.abs(Jw[d] - refs[d][0]).max() < 1e-10 * np.abs(refs[d][0]).max(), (nd, d)
                assert np.abs(Kw[d] - refs[d][1]).max() < 1e-10 * np.abs(refs[d][1]).max(), (nd, d)
        J1b, K1b = engine.fock_jk(P[0])                                     # bitwise reproducible, whatever ran in between
        assert np.array_equal(J1, J1b) and np.array_equal(K1, K1b)
    finally:
        _reset(engine)


def test_tiles_at_the_benched_size(engine):
    """N = 400: the tiles tensor element by element against the packed one on 20 000 samples; J and K against the packed layout's for one
    density and for the wide passes over four and eight densities (the matrix-core pass of `bench.py --layout tiles --n-dens 8`;
    scf:55-72, 27-44 for what J and K are -- the packed layout's are pinned to the reference at this size in test_gpu_parity.py)."""
    aos = _synthetic(400)
    N = 400
    rng = np.random.default_rng(5)
    P8 = np.stack([(lambda A: A + A.T)(rng.standard_normal((N, N))) for _ in range(8)])
    P = P8[0]
    idx = rng.integers(0, N, size=(20000, 4)).astype(np.int32)
    try:
        engine.set_basis(aos).build_eri(True, layout="packed")
        vp = engine.sample_eri(idx)
        Jp8, Kp8 = engine.fock_jk(P8)                                        # pairs of densities per pass
        Jp, Kp = Jp8[0], Kp8[0]
        engine.build_eri(True, layout="tiles")
        assert np.array_equal(engine.sample_eri(idx), vp)
        Jt, Kt = engine.fock_jk(P)
        assert np.abs(Jt - Jp).max() < 1e-11 * np.abs(Jp).max() and np.abs(Kt - Kp).max() < 1e-11 * np.abs(Kp).max()
        for nd in (4, 8):
            Jw, Kw = engine.fock_jk(P8[:nd])
            for d in range(nd):
                assert np.abs(Jw[d] - Jp8[d]).max() < 1e-11 * np.abs(Jp8[d]).max(), (nd, d)
                assert np.abs(Kw[d] - Kp8[d]).max() < 1e-11 * np.abs(Kp8[d]).max(), (nd, d)
    finally:
        _reset(engine)


def test_lockstep_batch_on_the_tiles_layout_takes_the_wide_pass(engine):
    """tf_scf_rhf_batch on a tiles tensor sends all densities of an iteration through ONE wide pass (up to eight as B-operand columns of
    the matrix-core products): the same energies as the batch on the packed tensor (pairs of densities per pass) and as single cycles,
    in a quarter of the passes (energy:315-540 for what the batch stands for; scf:1072-1154 for each cycle)."""
    import bench
    atoms, shells, aos, nocc, _ = bench.build_workload("synth-176")      # (the smallest of the series whose cycles converge from the core guess)
    try:
        xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
        res = {}
        for layout in ("packed", "tiles"):
            engine.set_basis(aos).build_eri(True, layout=layout)
            S, T, V, D, _ = engine.one_electron(xyz, chg, [0.0, 0.0, 0.5 * atoms[-1].origin[2]], spherical=True)
            X, _, _ = engine.orthogonaliser(S)
            _, C0 = engine.diagonalise(T + V, X)
            P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T
            P0 = 0.5 * (P0 + P0.T)
            E0 = float(np.sum(P0 * (T + V)))
            nao = [sum(s.n_sph for s in shells if s.atom == a) for a in range(2)]
            h = 0.002
            fields = [h * D[2], -h * D[2], h * D[0], -h * D[0], 2 * h * D[2], 2 * h * D[0]]
            kw = dict(X=X, conv="tight", damping="dynamic", n_atom_ao=nao, max_iter=200)
            rb = engine.scf_rhf_batch(S, T, V, [P0] * len(fields), [E0] * len(fields), nocc, mol.nuclear_repulsion(atoms), Fexts=fields, **kw)
            single = engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), Fext=fields[2], **kw)
            res[layout] = ([r["energy"] for r in rb], single["energy"], max(r["n_iter"] for r in rb), rb[0].get("passes"))
        Ep, Et = np.array(res["packed"][0]), np.array(res["tiles"][0])
        assert np.abs(Ep - Et).max() < 1e-9
        assert abs(res["tiles"][1] - Et[2]) < 1e-9 and abs(res["packed"][1] - Ep[2]) < 1e-9
        assert abs(Et[2] - Et[3]) < 1e-9                                      # E(+x) = E(-x)
    finally:
        _reset(engine)
