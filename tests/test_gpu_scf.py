"""GPU: the native RHF cycle (tf_scf_rhf: HIP J/K + rocBLAS + rocSOLVER) against the trajectories of the
reference's own tuna_scf.py (tests/golden) and the anchor energies of SURVEY.md section 6.2.  Bar: 1e-8 Eh."""
import numpy as np
import pytest

from conftest import atom_arrays, make_system
from oracle import scf_oracle as so
from tuna_amd import molecule as mol

pytestmark = pytest.mark.gpu

ANCHORS = {"h2_sto3g": -1.1167593075, "c2_n2_ccpvtz": -108.9834703056, "c4_co_def2tzvp": -112.7855372010,
           "c3_ar2_ccpvqz": -1053.6331483153}


def _prepare(engine, tag):
    atoms, shells, aos, nocc = make_system(tag)
    engine.set_basis(aos).build_eri(True)
    xyz, chg, org = atom_arrays(atoms)
    S, T, V, _, _ = engine.one_electron(xyz, chg, org, spherical=True)
    X, smallest, _ = engine.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    return atoms, S, T, V, X, P0, E0, nocc, ranges


@pytest.mark.parametrize("tag", ["h2_sto3g", "n2_sto3g", "he_631g", "n2_ccpvdz", "c2_n2_ccpvtz", "c4_co_def2tzvp", "c3_ar2_ccpvqz"])
def test_native_rhf_matches_reference_without_damping(engine, golden, small, tag):
    g = small[tag] if tag in small else golden(tag)
    atoms, S, T, V, X, P0, E0, nocc, ranges = _prepare(engine, tag)
    assert abs(E0 - float(g["E0"])) < 1e-9
    r = engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="none", n_atom_ao=ranges)
    ref_table = g["scf_table_nodamp"]
    assert r["converged"]
    assert abs(r["energy"] - float(g["scf_energy_nodamp"])) < 1e-9
    if tag in ANCHORS:
        assert abs(r["energy"] - ANCHORS[tag]) < 1e-8
    # the stopping iteration may differ by one when a criterion sits at its threshold to within rounding (1e-11 / 1e-12)
    n = min(r["n_iter"], len(ref_table))
    if tag == "he_631g":
        n = 3   # N = 2: the DIIS error vectors span one dimension, the Pulay system is singular to rounding and the
                # extrapolated iterates (step >= 3) are noise-driven in the reference too; only pre-DIIS steps compare
    else:
        assert abs(r["n_iter"] - len(ref_table)) <= 1
    np.testing.assert_allclose(r["table"][:n, 1], ref_table[:n, 1], atol=2e-9)        # E_total, every iteration
    np.testing.assert_allclose(r["table"][:n, 5], ref_table[:n, 5], atol=1e-8)        # commutator
    np.testing.assert_allclose(r["epsilons"], g["scf_eps_nodamp"], atol=1e-7)
    np.testing.assert_allclose(r["components"][:4], g["scf_components_nodamp"], atol=1e-7)
    # invariants of the converged state
    n_el = 2 * nocc
    assert abs(np.trace(r["P"] @ S) - n_el) < 1e-9
    assert np.abs(r["F"] @ r["P"] @ S - S @ r["P"] @ r["F"]).max() < 1e-7


@pytest.mark.parametrize("tag", ["h2_sto3g", "n2_sto3g", "n2_ccpvdz", "c4_co_def2tzvp", "c2_n2_ccpvtz", "c3_ar2_ccpvqz"])
def test_native_rhf_default_dynamic_damping(engine, golden, small, tag):
    """Default keywords (DIIS 6 + dynamic Zerner-Hehenberger damping): iteration count, per-iteration energies and damping
    factors of the reference run, including its always-zero "P_old_before_damping" (scf:1154 vs scf:1373)."""
    g = small[tag] if tag in small else golden(tag)
    atoms, S, T, V, X, P0, E0, nocc, ranges = _prepare(engine, tag)
    r = engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", n_atom_ao=ranges)
    assert abs(r["energy"] - float(g["scf_energy"])) < 1e-9
    ref = g["scf_table"]
    assert abs(r["n_iter"] - len(ref)) <= 1
    n = min(r["n_iter"], len(ref))
    np.testing.assert_allclose(r["table"][:n, 6], ref[:n, 6], atol=1e-6)               # damping factors
    np.testing.assert_allclose(r["table"][:n, 1], ref[:n, 1], atol=1e-8)


def test_oracle_loop_driven_by_hip_fock_builds(engine, golden):
    """The NumPy restatement of the reference loop with J/K coming from the HIP kernel (seam 2 of SURVEY.md section 8b)."""
    g = golden("n2_ccpvdz")
    atoms, S, T, V, X, P0, E0, nocc, ranges = _prepare(engine, "n2_ccpvdz")
    r = so.run_rhf(S, T, V, None, X, P0, E0, nocc, mol.nuclear_repulsion(atoms), ranges, conv="extreme", damping=False,
                   jk=engine.fock_jk)
    assert abs(r["energy"] - float(g["scf_energy_nodamp"])) < 1e-9
    assert abs(r["n_iter"] - len(g["scf_table_nodamp"])) <= 1


def test_medium_convergence_and_nonconvergence_error(engine):
    from tuna_amd._lib import TunaError
    atoms, S, T, V, X, P0, E0, nocc, ranges = _prepare(engine, "n2_sto3g")
    r = engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="medium", n_atom_ao=ranges)
    assert r["converged"] and abs(r["energy"] - (-106.76612847535482)) < 1e-6
    with pytest.raises(TunaError) as e:
        engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", max_iter=3, n_atom_ao=ranges)
    assert "not converged in 3 iterations" in str(e.value) and e.value.code == -4


def test_orthogonaliser(engine, golden):
    g = golden("c4_co_def2tzvp")
    S = so.to_spherical(g["U"], g["S"])
    X, smallest, Sinv = engine.orthogonaliser(S)
    Xo, so_small, Sio = so.orthogonaliser(S)
    assert np.abs(X - Xo).max() < 1e-10 and abs(smallest - so_small) < 1e-12 and np.abs(Sinv - Sio).max() < 1e-8 * np.abs(Sio).max()
    assert abs(smallest - float(g["smallest_S"])) < 1e-12


@pytest.mark.parametrize("tag", ["o2_triplet_sto3g", "o2_triplet_ccpvdz", "no_doublet_631g", "oh_doublet_ccpvdz", "li_doublet_631g"])
@pytest.mark.parametrize("damping", [True, False])
def test_unrestricted_cycle_matches_reference(engine, uhf_golden, tag, damping):
    """UHF (scf:1165-1281): fused two-density Fock builds on the GPU, trajectories of the reference's own unrestricted run."""
    from conftest import UHF_SYSTEMS
    from tuna_amd.energy import Calculation, build_molecule_and_integrals
    from tuna_amd.engine import SCF_CONVERGENCE
    from tuna_amd import scf
    g = uhf_golden[tag]
    sym, R, basis, na, nb = UHF_SYSTEMS[tag]
    calc = Calculation(basis=basis, SCF_conv=SCF_CONVERGENCE["extreme"], multiplicity=na - nb + 1, damping=damping, core_guess=True)
    molecule, integrals, X, guess, _ = build_molecule_and_integrals(sym, R, calc, engine)
    assert calc.reference == "UHF" and (molecule.n_alpha, molecule.n_beta) == (na, nb)
    assert abs(guess[3] - float(g["E0"])) < 1e-9
    out = scf.run_self_consistent_field_cycle(molecule, calc, integrals, float(g["V_NN"]), X, guess)
    sfx = "" if damping else "_nodamp"
    assert abs(out.energy - float(g["scf_energy" + sfx])) < 1e-9
    np.testing.assert_allclose(out.epsilons_alpha, g["eps_alpha" + sfx], atol=1e-7)
    np.testing.assert_allclose(out.epsilons_beta, g["eps_beta" + sfx], atol=1e-7)
    assert abs(np.trace(out.P_alpha @ integrals.S) - na) < 1e-9 and abs(np.trace(out.P_beta @ integrals.S) - nb) < 1e-9
    if damping and tag.startswith("o2"):
        return      # homonuclear UHF + dynamic damping is rounding-noise driven in the reference (see tests/test_oracle.py)
    ref = g["scf_table" + sfx]
    assert abs(out.n_iterations - len(ref)) <= 1
    n = min(out.n_iterations, len(ref))
    np.testing.assert_allclose(out.table[:n, 1], ref[:n, 1], atol=2e-8)
    np.testing.assert_allclose(out.table[:n, 6], ref[:n, 6], atol=1e-6)


def test_native_and_host_orchestrated_unrestricted_cycles_agree(engine, monkeypatch):
    """tf_scf_uhf (whole cycle in the library) against the host-orchestrated loop (the path a sharded tensor takes): same
    trajectory, same orbitals."""
    from conftest import UHF_SYSTEMS
    from tuna_amd.energy import Calculation, build_molecule_and_integrals
    from tuna_amd.engine import SCF_CONVERGENCE
    from tuna_amd import scf
    sym, R, basis, na, nb = UHF_SYSTEMS["oh_doublet_ccpvdz"]
    calc = Calculation(basis=basis, SCF_conv=SCF_CONVERGENCE["extreme"], multiplicity=na - nb + 1, damping=True, core_guess=True)
    molecule, integrals, X, guess, _ = build_molecule_and_integrals(sym, R, calc, engine)
    outs = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("TUNA_AMD_HOST_UHF", "1")
        outs.append(scf.run_self_consistent_field_cycle(molecule, calc, integrals, 0.0, X, guess))
    a, b = outs
    assert a.n_iterations == b.n_iterations and abs(a.energy - b.energy) < 1e-10
    np.testing.assert_allclose(a.table[:, 1:], b.table[:, 1:], atol=1e-8)
    np.testing.assert_allclose(a.epsilons_alpha, b.epsilons_alpha, atol=1e-8)
    np.testing.assert_allclose(a.epsilons_beta, b.epsilons_beta, atol=1e-8)
    assert np.abs(a.P_alpha - b.P_alpha).max() < 1e-8 and np.abs(a.P_beta - b.P_beta).max() < 1e-8
    assert np.abs(a.F_alpha - b.F_alpha).max() < 1e-8


def test_uhf_input_line(uhf_golden):
    from tuna_amd.energy import run
    out = run("SPE : O O 1.2075 : UHF CC-PVDZ : EXTREME NODAMP ML 3 COREGUESS")
    assert abs(out.energy - float(uhf_golden["o2_triplet_ccpvdz"]["scf_energy_nodamp"])) < 1e-8


@pytest.mark.parametrize("symbols,R,basis,nocc", [(["C", "O"], 1.128, "cc-pVQZ", 7), (["AR", "AR"], 3.76, "cc-pVQZ", 18),
                                                 (["F", "H"], 0.917, "aug-cc-pVQZ", 5)])
def test_eigenvector_refinement_matches_the_exact_eigensolver(symbols, R, basis, nocc):
    """n > 64: inside the cycle the density comes from refined eigenvectors (tf_scf.hip.h); the same run with every
    diagonalisation done by rocsolver_dsyevd (TF_EIGH=rocsolver) must give the same table, orbitals and energy."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
from oracle import scf_oracle as so
symbols, R, basis, nocc = %r, %r, %r, %r
atoms = mol.make_atoms(symbols, mol.angstrom_to_bohr(R)); sh = mol.build_shells(atoms, basis); aos = mol.expand_cartesian_aos(sh)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    S, T, V, _, _ = eng.one_electron([a.origin for a in atoms], [float(a.charge) for a in atoms], [0, 0, 0.5 * atoms[1].origin[2]])
    X, _, _ = eng.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    nao = [sum(s.n_sph for s in sh if s.atom == a) for a in range(2)]
    r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", n_atom_ao=nao)
    C = r["C"]; proj = C[:, :nocc] @ C[:, :nocc].T
    print(json.dumps({"N": eng.N, "E": r["energy"], "n_iter": r["n_iter"], "table": np.asarray(r["table"])[:r["n_iter"]].tolist(),
                      "eps": np.asarray(r["epsilons"]).tolist(), "proj_norm": float(np.abs(proj).sum()), "P_sum": float(np.abs(r["P"]).sum()),
                      "orth": float(np.abs(C.T @ S @ C - np.eye(eng.N)).max())}))
''' % (os.path.join(os.path.dirname(__file__), ".."), symbols, R, basis, nocc)
    res = {}
    for mode in ("refine", "rocsolver"):
        env = dict(os.environ)
        env.pop("TF_EIGH", None)
        if mode == "rocsolver":
            env["TF_EIGH"] = "rocsolver"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    a, b = res["refine"], res["rocsolver"]
    assert a["N"] > 64
    assert abs(a["E"] - b["E"]) < 1e-9
    # the last iterations sit at the EXTREME thresholds (max dP ~ 1e-11) within rounding noise: the stopping iteration may differ
    assert abs(a["n_iter"] - b["n_iter"]) <= 3
    n = min(a["n_iter"], b["n_iter"])
    ta, tb = np.asarray(a["table"])[:n], np.asarray(b["table"])[:n]
    assert np.abs(ta[:, 1] - tb[:, 1]).max() < 1e-8                       # energies of every iteration
    big = tb[:, 5] > 1e-9
    assert np.abs(ta[big, 5] / tb[big, 5] - 1.0).max() < 1e-3              # commutator norms while they are above the noise
    assert np.abs(ta[big, 6] - tb[big, 6]).max() < 1e-6                    # same damping factors
    assert np.abs(np.asarray(a["eps"]) - np.asarray(b["eps"])).max() < 1e-8   # final orbital energies come from a real eigensolve
    assert a["orth"] < 1e-10 and abs(a["P_sum"] - b["P_sum"]) < 1e-6 * b["P_sum"]


@pytest.mark.parametrize("reference", ["RHF", "UHF"])
def test_electric_field_term_of_the_fock_matrix(engine, reference):
    """The field term of the Fock matrix (F_fld in scf:525, `integrals.F`) through the native cycles: the field energy component is
    tr(P F), and the energy responds to a small field with the dipole expectation value, dE/d(eps) = tr(P D_z) (Hellmann-Feynman for
    the variational SCF energy): central difference against the analytic value."""
    from tuna_amd.energy import Calculation, build_molecule_and_integrals
    from tuna_amd.engine import SCF_CONVERGENCE
    from tuna_amd import scf
    mult = 1 if reference == "RHF" else 3
    sym = ["H", "F"] if reference == "RHF" else ["N", "H"]
    calc = Calculation(basis="6-31G", SCF_conv=SCF_CONVERGENCE["extreme"], multiplicity=mult, damping=False, core_guess=True)
    molecule, integrals, X, guess, _ = build_molecule_and_integrals(sym, mol.angstrom_to_bohr(0.95), calc, engine)
    assert calc.reference == reference
    Dz = np.asarray(integrals.D)[2]
    eps = 2e-4
    res = {}
    for f in (-eps, 0.0, eps):
        integrals.F = f * Dz
        out = scf.run_self_consistent_field_cycle(molecule, calc, integrals, 0.0, X, guess)
        assert abs(out.electric_field_energy - np.sum(out.P * integrals.F)) < 1e-11
        res[f] = out
    slope = (res[eps].energy - res[-eps].energy) / (2 * eps)
    mu = float(np.sum(res[0.0].P * Dz))
    assert abs(mu) > 1e-2                                           # (electronic dipole about the centre of mass: not zero by symmetry)
    assert abs(slope - mu) < 2e-6


def test_symmetry_blocked_eigensolve_and_its_refusal():
    """tf_scf.hip.h: eigh_blocked -- the x/y parity classes of a diatomic are solved as one batch of small problems when (and only when)
    the matrix has no element connecting two classes.  scf:222-250 for both cases: the same orbital energies as LAPACK on the full matrix,
    S-orthonormal orbitals that solve F C = S C eps; a dipole field along x couples the classes and must take the full solve."""
    from conftest import atom_arrays, make_system
    from tuna_amd.engine import Engine
    atoms, shells, aos, nocc = make_system("c3_ar2_ccpvqz")
    with Engine(0) as eng:
        eng.set_basis(aos).build_eri(True)
        N = eng.N
        assert N > 64
        xyz, chg, org = atom_arrays(atoms)
        S, T, V, D, _ = eng.one_electron(xyz, chg, org, spherical=True)
        s0 = eng.eigh_stats()
        X, smin, _ = eng.orthogonaliser(S)
        s1 = eng.eigh_stats()
        assert s1["blocked_solves"] == s0["blocked_solves"] + 1            # S itself is block diagonal
        assert np.abs(X @ S @ X - np.eye(N)).max() < 1e-10
        F = T + V

        def check(Fm):
            eps, Cm = eng.diagonalise(Fm, X)
            ref = np.linalg.eigvalsh(0.5 * (X.T @ Fm @ X + (X.T @ Fm @ X).T))
            # (as accurate as a LAPACK solve of the full matrix: n x machine epsilon x norm.  A padding diagonal far above the spectrum
            # once cost the blocked solver three digits of that -- enough to mix near-degenerate g/u pairs)
            assert np.abs(eps - ref).max() < 1e-13 * N * max(1.0, np.abs(ref).max())
            assert np.all(np.diff(eps) >= 0)
            assert np.abs(Cm.T @ S @ Cm - np.eye(N)).max() < 1e-10
            assert np.abs(Fm @ Cm - S @ Cm * eps).max() < 1e-8 * max(1.0, np.abs(ref).max())
        check(F)
        s2 = eng.eigh_stats()
        assert s2["blocked_solves"] == s1["blocked_solves"] + 1 and s2["blocked_declined"] == s1["blocked_declined"]
        check(F + 0.01 * D[2])                                             # a field along z keeps the classes apart
        s3 = eng.eigh_stats()
        assert s3["blocked_solves"] == s2["blocked_solves"] + 1
        check(F + 0.01 * D[0])                                             # a field along x does not: full solve
        s4 = eng.eigh_stats()
        assert s4["blocked_solves"] == s3["blocked_solves"] and s4["blocked_declined"] == s3["blocked_declined"] + 1


@pytest.mark.parametrize("kind", ["rhf", "uhf", "mixed"])
def test_class_diagonal_task_list_gives_the_same_bits(kind):
    """tf_device.hip: launch_jk_packed -- the native cycles send class-diagonal densities (no element between AOs of different x/y parity)
    over a shorter task list; the skipped products are exact zeros, so the whole cycle (scf:1072-1154 / 1165-1281) must come out bit for
    bit as with the full list (TF_JK_CLASS_DIAGONAL=0), and the counters must show that the shorter list was really taken."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from conftest import atom_arrays, make_system, make_uhf_system
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
from oracle import scf_oracle as so
kind = %r
with Engine(0) as eng:
    if kind in ("rhf", "mixed"):
        atoms, shells, aos, nocc = make_system("c3_ar2_ccpvqz")
    else:
        atoms, shells, aos, na, nb = make_uhf_system("o2_triplet_ccpvdz")
    eng.set_basis(aos).build_eri(True)
    xyz, chg, org = atom_arrays(atoms)
    S, T, V, _, _ = eng.one_electron(xyz, chg, org, spherical=True)
    X, _, _ = eng.orthogonaliser(S)
    nao = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    if kind == "mixed":
        # a two-density cycle, then a one-density cycle on the same context: the partial sums of the two pass types are laid out
        # differently, and the slots a skipped task leaves alone must not hold what the other pass type wrote there
        _, C0 = eng.diagonalise(T + V, X)
        Ph = C0[:, :nocc] @ C0[:, :nocc].T; Ph = 0.5 * (Ph + Ph.T)
        ru = eng.scf_uhf(S, T, V, Ph, Ph, float(np.sum(2 * Ph * (T + V))), nocc, nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping="dynamic", n_atom_ao=nao)
        r = eng.scf_rhf(S, T, V, 2 * Ph, float(np.sum(2 * Ph * (T + V))), nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping="dynamic", n_atom_ao=nao)
        assert abs(ru["energy"] - r["energy"]) < 1e-8
        mats = [r["P"], r["F"], ru["P_spin"][0]]
    elif kind == "rhf":
        _, C0 = eng.diagonalise(T + V, X)
        P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T; P0 = 0.5 * (P0 + P0.T)
        r = eng.scf_rhf(S, T, V, P0, float(np.sum(P0 * (T + V))), nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping="dynamic", n_atom_ao=nao)
        mats = [r["P"], r["F"]]
    else:
        _, C0 = eng.diagonalise(T + V, X)
        Pa = C0[:, :na] @ C0[:, :na].T; Pb = C0[:, :nb] @ C0[:, :nb].T
        E0 = float(np.sum((Pa + Pb) * (T + V)))
        r = eng.scf_uhf(S, T, V, 0.5 * (Pa + Pa.T), 0.5 * (Pb + Pb.T), E0, na, nb, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping="dynamic", n_atom_ao=nao)
        mats = [r["P_spin"][0], r["P_spin"][1]]
    print(json.dumps({"E": float(r["energy"]).hex(), "n_iter": int(r["n_iter"]), "table": [float(x).hex() for x in np.asarray(r["table"])[:r["n_iter"], 1]],
                      "sums": [float(np.abs(m).sum()).hex() for m in mats], "paths": eng.jk_path_stats()}))
''' % (os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__), kind)
    res = {}
    for mode in ("short", "full"):
        env = dict(os.environ)
        env.pop("TF_JK_CLASS_DIAGONAL", None)
        env["TF_JK_CD_NMIN"] = "0"                                   # (by default only tensors of N >= 160 take the test: below, it costs what it saves)
        if mode == "full":
            env["TF_JK_CLASS_DIAGONAL"] = "0"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    a, b = res["short"], res["full"]
    assert a["paths"]["class_diagonal_passes"] >= a["n_iter"] - 2 and b["paths"]["class_diagonal_passes"] == 0
    assert a["E"] == b["E"] and a["n_iter"] == b["n_iter"] and a["table"] == b["table"] and a["sums"] == b["sums"]


def test_mixed_sequence_on_one_context_equals_the_plain_paths():
    """A context used the way a finite-field property run uses it -- restricted cycle, lockstep batch with fields along z and x (pairs of
    densities per pass), restricted cycle with a field, unrestricted cycle, a Fock build through the public entry -- at a size where the
    class-diagonal task list, the blocked eigensolver and the block labels of the refinement are all active (synthetic 176-AO Ar2-like
    system; scf:1072-1281 for the cycles, energy:315-540 for the batch).  Every energy must equal the one of the same sequence with those
    paths switched off (TF_JK_CLASS_DIAGONAL=0 TF_EIGH_BLOCKS=0) to 1e-9: nothing may leak from one call into the next."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import bench
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
atoms, shells, aos, nocc, desc = bench.build_workload("synth-176")
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    N = eng.N
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, D, Q = eng.one_electron(xyz, chg, [0.0, 0.0, 0.5 * atoms[-1].origin[2]], spherical=True)
    X, _, _ = eng.orthogonaliser(S)
    _, C0 = eng.diagonalise(T + V, X)
    P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T; P0 = 0.5 * (P0 + P0.T)
    E0 = float(np.sum(P0 * (T + V)))
    nao = [sum(s.n_sph for s in shells if s.atom == a) for a in range(2)]
    VNN = mol.nuclear_repulsion(atoms)
    kw = dict(X=X, conv="tight", damping="dynamic", n_atom_ao=nao, max_iter=200)
    out = {}
    out["rhf"] = eng.scf_rhf(S, T, V, P0, E0, nocc, VNN, **kw)["energy"]
    h = 0.002
    fields = [h * D[2], -h * D[2], h * D[0], -h * D[0]]
    rb = eng.scf_rhf_batch(S, T, V, [P0] * 4, [E0] * 4, nocc, VNN, Fexts=fields, **kw)
    out["batch"] = [r["energy"] for r in rb]
    out["rhf_z"] = eng.scf_rhf(S, T, V, P0, E0, nocc, VNN, Fext=fields[0], **kw)["energy"]
    out["rhf_x"] = eng.scf_rhf(S, T, V, P0, E0, nocc, VNN, Fext=fields[2], **kw)["energy"]
    out["uhf"] = eng.scf_uhf(S, T, V, P0 / 2, P0 / 2, E0, nocc, nocc, VNN, **kw)["energy"]
    out["rhf_again"] = eng.scf_rhf(S, T, V, P0, E0, nocc, VNN, **kw)["energy"]
    A = np.random.default_rng(5).standard_normal((N, N)); Pr = A + A.T
    J, K = eng.fock_jk(Pr[None])
    out["jk"] = [float(np.abs(J).sum()), float(np.abs(K).sum())]
    out["paths"] = eng.jk_path_stats(); out["eigh"] = eng.eigh_stats(); out["N"] = N
    print(json.dumps(out))
''' % (os.path.join(os.path.dirname(__file__), ".."),)
    res = {}
    for mode in ("round4", "plain"):
        env = dict(os.environ)
        for k in ("TF_JK_CLASS_DIAGONAL", "TF_EIGH_BLOCKS"):
            env.pop(k, None)
        if mode == "plain":
            env["TF_JK_CLASS_DIAGONAL"] = "0"; env["TF_EIGH_BLOCKS"] = "0"
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert o.returncode == 0, o.stderr[-2000:]
        res[mode] = json.loads(o.stdout.strip().splitlines()[-1])
    a, b = res["round4"], res["plain"]
    assert a["N"] >= 160 and a["paths"]["class_diagonal_passes"] > 20 and a["eigh"]["blocked_solves"] > 5
    assert b["paths"]["class_diagonal_passes"] == 0 and b["eigh"]["blocked_solves"] == 0
    for key in ("rhf", "rhf_z", "rhf_x", "uhf", "rhf_again"):
        assert abs(a[key] - b[key]) < 1e-9, (key, a[key], b[key])
    assert np.abs(np.array(a["batch"]) - np.array(b["batch"])).max() < 1e-9
    assert abs(a["rhf"] - a["rhf_again"]) < 1e-10 and abs(a["uhf"] - a["rhf"]) < 1e-8
    assert abs(a["batch"][0] - a["rhf_z"]) < 1e-9 and abs(a["batch"][2] - a["rhf_x"]) < 1e-9 and abs(a["batch"][2] - a["batch"][3]) < 1e-9
    assert np.allclose(a["jk"], b["jk"], rtol=1e-13, atol=0)


def test_blocked_refinement_of_an_unrestricted_cycle():
    """64 < n with parity blocks <= 64 (NO doublet / cc-pVQZ, N = 110): the densities of both spins come from the in-LDS refinement kernel
    run on all blocks in one launch (tf_scf.hip.h: ref_refine_blocks, one slot per spin); the cycle (scf:1165-1281) must reproduce the
    run with the GEMM refinement at the full dimension (TF_REFINE_BLOCKS=0) and the one with every solve exact (TF_EIGH=rocsolver)."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json
sys.path.insert(0, %r)
from tuna_amd.energy import run
out = run("SPE : N O 1.151 : UHF CC-PVQZ : ML 2")
print(json.dumps({"E": out.energy, "n": out.n_iterations}))
''' % (os.path.join(os.path.dirname(__file__), ".."),)
    res = {}
    for mode, var in (("blocks", None), ("gemm", ("TF_REFINE_BLOCKS", "0")), ("exact", ("TF_EIGH", "rocsolver"))):
        env = dict(os.environ)
        env.pop("TF_REFINE_BLOCKS", None); env.pop("TF_EIGH", None)
        if var:
            env[var[0]] = var[1]
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert o.returncode == 0, o.stderr[-2000:]
        res[mode] = json.loads(o.stdout.strip().splitlines()[-1])
    assert abs(res["blocks"]["E"] - res["exact"]["E"]) < 1e-9 and abs(res["gemm"]["E"] - res["exact"]["E"]) < 1e-9
    assert abs(res["blocks"]["n"] - res["exact"]["n"]) <= 2
