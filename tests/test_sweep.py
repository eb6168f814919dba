"""Thirty closed-shell systems across H .. Ar and all eighteen basis sets the reference ships (tests/golden/sweep_systems.json: one
run of the reference's own RHF cycle each -- tools/make_golden.py --sweep-only): integrals and the SCF of the HIP path, and the
oracle's integrals on the smaller ones."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, atom_arrays
from oracle import oracle as orc
from oracle import scf_oracle as so
from tuna_amd import molecule as mol

SWEEP = json.load(open(os.path.join(GOLD, "sweep_systems.json")))


def _system(g):
    R = None if g["R_angstrom"] is None else mol.angstrom_to_bohr(g["R_angstrom"])
    atoms = mol.make_atoms(g["symbols"], R)
    shells = mol.build_shells(atoms, g["basis"])
    return atoms, shells, mol.expand_cartesian_aos(shells)


def test_every_basis_set_is_covered():
    norm = lambda s: s.upper().replace("-", "_").replace("*", "STAR").lstrip("_")
    assert {norm(b) for b in mol.available_basis_sets()} <= {norm(g["basis"]) for g in SWEEP.values()}


@pytest.mark.parametrize("tag", [t for t, g in SWEEP.items() if g["n_ao"] <= 20])
def test_oracle_integrals_against_the_reference(tag):
    """CPU: the C restatement reproduces the reference's integral matrices for the small members of the sweep."""
    g = SWEEP[tag]
    atoms, shells, aos = _system(g)
    from tuna_amd import spherical
    U = spherical.transformation_matrix([s.L for s in shells])
    xyz, chg, org = atom_arrays(atoms)
    S, T, V, _, _ = orc.one_electron(aos, xyz, chg, org)
    for M, name in ((S, "S_fro"), (T, "T_fro"), (V, "V_fro")):
        assert abs(np.linalg.norm(U @ M @ U.T) - g[name]) < 1e-11 * max(1.0, g[name]), name
    Es = so.eri_to_spherical(U, orc.eri(aos))
    assert abs(np.sqrt(np.sum(Es * Es)) - g["eri_fro"]) < 1e-11 * max(1.0, g["eri_fro"])
    idx = np.array(g["eri_idx"])
    assert np.abs(Es[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] - np.array(g["eri_val"])).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(SWEEP))
def test_hip_path_against_the_reference(engine, tag):
    """GPU: integrals, the tensor and the native RHF cycle (core guess, EXTREME, dynamic damping) of every system of the sweep
    against the reference's run: energy 1e-8 Eh, orbital energies 1e-6, iteration count within one."""
    g = SWEEP[tag]
    atoms, shells, aos = _system(g)
    nocc = g["n_occ"]
    engine.set_basis(aos).build_eri(True)
    assert engine.N == g["n_ao"]
    xyz, chg, org = atom_arrays(atoms)
    S, T, V, _, _ = engine.one_electron(xyz, chg, org, spherical=True)
    for M, name in ((S, "S_fro"), (T, "T_fro"), (V, "V_fro")):
        assert abs(np.linalg.norm(M) - g[name]) < 1e-10 * max(1.0, g[name]), name
    idx = np.array(g["eri_idx"], dtype=np.int32)
    assert np.abs(engine.sample_eri(idx) - np.array(g["eri_val"])).max() < 1e-11
    X, _, _ = engine.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    assert abs(E0 - g["E0"]) < 1e-8
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    V_NN = mol.nuclear_repulsion(atoms)
    assert abs(V_NN - g["V_NN"]) < 1e-10
    r = engine.scf_rhf(S, T, V, P0, E0, nocc, V_NN, X=X, conv="extreme", damping="dynamic", n_atom_ao=ranges)
    assert r["converged"]
    assert abs(r["energy"] - g["energy"]) < 1e-8
    # EXTREME asks for |dE| < 1e-12 on energies of up to 1e3 Eh: the last iterations sit at the rounding floor of the energy sum, where the
    # summation order decides when the test first passes (LiF/6-31G: same trajectory to 1e-10 for 22 iterations, then two more)
    assert abs(r["n_iter"] - g["n_iter"]) <= 2
    np.testing.assert_allclose(r["epsilons"], np.array(g["epsilons"]), atol=1e-6)
    np.testing.assert_allclose(r["components"][:4], np.array(g["components"]), atol=1e-6)


UHF_SWEEP = json.load(open(os.path.join(GOLD, "uhf_sweep.json")))
MP2_SWEEP = json.load(open(os.path.join(GOLD, "mp2_sweep.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(UHF_SWEEP))
def test_open_shell_sweep_against_the_reference(engine, tag):
    """GPU: the native unrestricted cycle (tf_scf_uhf: both spin densities in one pass over the tensor) on atoms and diatomic radicals
    / cations across the basis sets, against one run of the reference's own UHF cycle each (core guess, EXTREME, no damping)."""
    from tuna_amd import scf
    from tuna_amd.energy import Calculation, build_molecule_and_integrals
    from tuna_amd.engine import SCF_CONVERGENCE
    g = UHF_SWEEP[tag]
    na, nb = g["n_alpha"], g["n_beta"]
    R = None if g["R_angstrom"] is None else mol.angstrom_to_bohr(g["R_angstrom"])
    charge = sum(a.charge for a in mol.make_atoms(g["symbols"], R)) - (na + nb)
    calc = Calculation(basis=g["basis"], SCF_conv=SCF_CONVERGENCE["extreme"], multiplicity=na - nb + 1, damping=False, core_guess=True,
                       charge=int(charge))
    molecule, integrals, X, guess, _ = build_molecule_and_integrals(g["symbols"], R, calc, engine)
    assert calc.reference == "UHF" and (molecule.n_alpha, molecule.n_beta) == (na, nb) and engine.N == g["n_ao"]
    assert abs(guess[3] - g["E0"]) < 1e-8
    out = scf.run_self_consistent_field_cycle(molecule, calc, integrals, g["V_NN"], X, guess)
    assert abs(out.energy - g["energy"]) < 1e-8
    assert abs(out.n_iterations - g["n_iter"]) <= 3          # (the last iterations of an EXTREME run sit at the rounding floor: see above)
    np.testing.assert_allclose(out.epsilons_alpha, np.array(g["eps_alpha"]), atol=1e-6)
    if nb > 0:
        np.testing.assert_allclose(out.epsilons_beta, np.array(g["eps_beta"]), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(MP2_SWEEP))
def test_mp2_sweep_against_the_reference(engine, tag):
    """GPU: RMP2 (AO->MO of the (ia|jb) block on the packed rows + the energy sums) on the reference's converged orbitals of twelve
    members of the sweep, against the reference's own transformation and energy expressions: 1e-10 Eh per component."""
    g = MP2_SWEEP[tag]
    atoms, shells, aos = _system(g)
    engine.set_basis(aos).build_eri(True)
    r = engine.mp2_rhf(np.array(g["C"]), np.array(g["eps"]), g["n_occ"])
    assert abs(r["E_OS"] - g["E_OS"]) < 1e-10 and abs(r["E_SS"] - g["E_SS"]) < 1e-10
